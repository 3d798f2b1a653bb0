// phi_solve.hip -- stages 2b-3 of ILP_function (src/ILP_index.cpp:670-1525) behind phi_solve().
//
// Host orchestration of GPU kernels:
//   1. match + shared-anchor filter                      (anchors.hip)    <- :643-743
//   2. exact solve replacing model.optimize() (:1418):   (dp.hip)
//        The reference objective counts each minimiser at most once (z_i, :830/:876):
//            true(P) = #{i : some anchor of i traversed by P} - 2*(R/2) * #recombinations(P).
//        The DP maximises an ADDITIVE score sum_a w_a [a traversed] - cost.  Used three ways:
//          upper bound   for any set S of minimisers:  true(P) <= |S| + DP(w = 1 outside S, 0 on S)
//          lower bound   true(P) of any path the DP returns
//          branching     OPT = max over "minimiser i is counted only at cluster c of its anchors"
//                        (clusters = groups of pairwise mutually exclusive anchors), each child
//                        again solved by the DP with the other clusters' weights set to 0.
//        The root closes with one DP run when the optimal path traverses no minimiser twice and
//        typically with two or three otherwise (S := the doubly-counted minimisers).
//   3. decode (:1431-1525): path vertices / haplotype labels, recombination count.
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <set>
#include <unordered_map>
#include "phi_ctx.h"
#include "phi_dev.h"

#define HIPCHK(call) do { int rc_ = phi_hip_check(c, (call), #call); if (rc_) return rc_; } while (0)
#define PHICHK(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

enum { S_ERR = 0, S_NBAD = 1, S_BATCHBAD = 2, S_NEMIT = 3, S_FILTERED = 4, S_INMODEL = 5, S_EXPORT = 6, S_BATCHBAD2 = 7,
       S_OVCNT = 8 /* .. 10: three rotating overflow counters (phi_ctx.h) */, S_N = 11 };
static uint64_t *scalar(phi_ctx *c, int i) { return c->d_scalars.as<uint64_t>() + i; }

// The walk entries as the solve's host code reads them: from the context's host copy, or -- a chromosome-scale graph keeps none
// (phi_set_graph) -- from the device copy: single entries for the backtrack, whole stretches for the decoded path.
static bool have_host_walks(const phi_ctx *c) { return (int64_t)c->h_walk_vtx.size() == c->n_entries; }
static int walk_vtx_at(phi_ctx *c, int64_t e, int32_t *v)
{
    if (have_host_walks(c)) { *v = c->h_walk_vtx[(size_t)e]; return PHI_OK; }
    return phi_hip_check(c, phi_copy_sync(c, v, c->d_walk_vtx.as<int32_t>() + e, 4, hipMemcpyDeviceToHost), "walk entry D2H");
}
static int walk_vtx_range(phi_ctx *c, int64_t es, int64_t n, int32_t *dst)
{
    if (have_host_walks(c)) { memcpy(dst, c->h_walk_vtx.data() + es, (size_t)n * 4); return PHI_OK; }
    return phi_hip_check(c, phi_copy_sync(c, dst, c->d_walk_vtx.as<int32_t>() + es, (size_t)n * 4, hipMemcpyDeviceToHost), "walk entries D2H");
}
// (the branch and bound proper and the host-side score bound index the array freely: fetched whole, once)
static int ensure_host_walks(phi_ctx *c)
{
    if (have_host_walks(c)) return PHI_OK;
    if (!c->h_walk_vtx.resize((size_t)c->n_entries)) return phi_fail(c, PHI_ERR_NOMEM, "host allocation failed");
    return phi_hip_check(c, phi_copy_sync(c, c->h_walk_vtx.data(), c->d_walk_vtx.p, (size_t)c->n_entries * 4, hipMemcpyDeviceToHost), "walk entries D2H");
}
static uint64_t pow2_at_least(uint64_t x) { uint64_t p = 1; while (p < x) p <<= 1; return p; }

struct Seg { int32_t h; int64_t es, ee; };     // path segment: walk h, entries es..ee (inclusive)

// flags[n] -> ascending indices in out; *n_out = count
int phi_compact(phi_ctx *c, const uint8_t *flags, int64_t n, DevBuf &out, int64_t *n_out)
{
    *n_out = 0;
    const int64_t nb = phi_compact_num_blocks(n);
    if (nb == 0) return PHI_OK;
    PHICHK(phi_dev_ensure(c, c->d_blk_cnt, (size_t)nb * 4));
    PHICHK(phi_dev_ensure(c, c->d_blk_off, (size_t)(nb + 1) * 8));
    phi_launch_flag_count(c->stream, flags, n, c->d_blk_cnt.as<int32_t>());
    PHICHK(phi_scan_counts_wide(c, c->d_blk_cnt.as<int32_t>(), nb, c->d_blk_off.as<int64_t>()));
    int64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, c->d_blk_off.as<int64_t>() + nb, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    PHICHK(phi_dev_ensure(c, out, (size_t)std::max<int64_t>(total, 1) * 4));
    phi_launch_flag_write(c->stream, flags, n, c->d_blk_off.as<int64_t>(), out.as<int32_t>());
    *n_out = total;
    return PHI_OK;
}

// the host copy of the kept anchors (a large model is solved on the device copy: solve_dev.hip)
int phi_host_anchors(phi_ctx *c)
{
    if (c->anchors_host) return PHI_OK;
    const int64_t n = c->n_kept;
    PHICHK(phi_pin_ensure(c, (size_t)std::max<int64_t>(n, 1) * sizeof(PhiAnchorHost)));
    c->h_kept = PhiAnchorSpan{static_cast<PhiAnchorHost *>(c->h_pin), n};
    if (n) HIPCHK(hipMemcpyAsync(c->h_kept.p, c->d_anchors.p, (size_t)n * 12, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->h_dp_own.clear();
    c->h_dp = c->h_kept;                                       // (device mode only runs when every kept anchor spans an edge)
    c->anchors_host = true;
    return PHI_OK;
}

// ------------------------------------------------------------------------------------------ blocks of steps
// Up to 64 walks: cut the chain of compact steps into blocks that the DP kernel solves in parallel (dp_events.hip
// DP_ROW / DP_PATH).  A cut may sit before step k when the graph allows it (phi_set_graph: no recombination edge from
// before the cut, not inside a pair of allele steps) and, on every walk, some entry between the walk's events around the
// cut is split by no dp anchor -- found here, once per solve, from all dp anchors whatever their weights.
static int dp_prepare_blocks(phi_ctx *c, int64_t n_dp)
{
    c->dp_blocks = false;
    c->n_blk = 0;
    c->dp_cls = false;
    if (!c->dp_events || c->n_walks > PHI_DP_EVENT_MAX_WALKS || getenv("PHI_DP_NOBLOCKS") || c->n_k < 4) return PHI_OK;
    // more than 64 walks: short blocks, whose rows run on class lanes (dp_events.hip); few sites per block keep the
    // walks of a block in at most 64 classes
    const bool cls = c->n_walks > 64;
    c->blk_ls = cls ? 256 : 64;
    const int64_t ne = c->n_entries;
    const int32_t nk = c->n_k;
    // (scratch of ne + 3 ints twice: the per-run prefix-sum buffers, which no run is using yet -- at chromosome scale
    //  every buffer of this size is 5 GB that the driver clears on allocation)
    PHICHK(phi_dev_ensure(c, c->d_off_end, (size_t)(ne + 3) * 4));
    PHICHK(phi_dev_ensure(c, c->d_off_start, (size_t)(ne + 3) * 4));
    PHICHK(phi_dev_ensure(c, c->d_stepdiff, (size_t)(nk + 2) * 4));
    {
        const int64_t nb = phi_scan_i32_num_blocks(ne + 2);
        PHICHK(phi_dev_ensure(c, c->d_scan_blk, (size_t)nb * 4));
        PHICHK(phi_dev_ensure(c, c->d_scan_blkoff, (size_t)(nb + 1) * 8));
    }
    int32_t *d_a = c->d_off_end.as<int32_t>(), *d_b = c->d_off_start.as<int32_t>();
    HIPCHK(hipMemsetAsync(c->d_stepdiff.p, 0, (size_t)(nk + 2) * 4, c->stream));
    if (getenv("PHI_CUT_COUNTED")) {                                   // (tests: the flags from counted coverage, as until round 4)
        HIPCHK(hipMemsetAsync(d_a, 0, (size_t)(ne + 3) * 4, c->stream));
        phi_launch_cut_cov(c->stream, c->d_a_e1.as<phi_ent_t>(), c->d_g_span.as<uint8_t>(), n_dp, d_a);
        phi_launch_scan_i32(c->stream, d_a, ne + 1, d_b, c->d_scan_blk.as<int32_t>(), c->d_scan_blkoff.as<int64_t>());     // d_b[e + 1] = anchors a cut before e splits
        phi_launch_cut_clean(c->stream, d_b, ne, d_a);                                                                       // d_a[e] = clean
    } else {
        phi_launch_cut_clean_direct(c->stream, c->d_g_off.as<int64_t>(), c->d_g_span.as<uint8_t>(), ne, d_a);               // d_a[e] = clean
    }
    phi_launch_scan_i32(c->stream, d_a, ne + 1, d_b, c->d_scan_blk.as<int32_t>(), c->d_scan_blkoff.as<int64_t>());     // d_b[e] = clean entries before e
    phi_launch_cut_events(c->stream, c->d_ev_e.as<phi_ent_t>(), c->n_ev, c->d_ev_off.as<int64_t>(), c->d_walk_off.as<int64_t>(), c->n_walks,
                          c->d_walk_vtx.as<int32_t>(), c->d_cvtx.as<int32_t>(), d_b, c->d_stepdiff.as<int32_t>());
    std::vector<int32_t> closed((size_t)nk + 2);
    HIPCHK(hipMemcpyAsync(closed.data(), c->d_stepdiff.p, (size_t)(nk + 2) * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int32_t k = 1; k <= nk + 1; k++) closed[(size_t)k] += closed[(size_t)k - 1];          // > 0: some walk forbids a cut before step k
    auto cut_ok = [&](int32_t k) { return k > 0 && k < nk && c->h_k_cut_ok[(size_t)k] && closed[(size_t)k] == 0; };
    if (getenv("PHI_TIMING")) {
        int64_t n_struct = 0, n_both = 0, longest = 0, run = 0, longest_s = 0, run_s = 0;
        for (int32_t k = 1; k < nk; k++) {
            n_struct += c->h_k_cut_ok[(size_t)k] != 0;
            if (cut_ok(k)) { n_both++; run = 0; } else longest = std::max(longest, ++run);
            if (c->h_k_cut_ok[(size_t)k]) run_s = 0; else longest_s = std::max(longest_s, ++run_s);
        }
        fprintf(stderr, "[phi timing] solve: cuts: %lld of %d steps allowed by the graph (longest closed stretch %lld), %lld also clean of anchors (longest closed stretch %lld)\n",
                (long long)n_struct, nk, (long long)longest_s, (long long)n_both, (long long)longest);
    }
    // block length: enough blocks to fill the machine with (walks + 1) tasks each, few enough to keep the chain short;
    // a block is at most as long as its ring of tops: 1024 steps, or 2048 where the longest stretch without a cut needs it
    // (256-step blocks first: their tasks take 51 instead of 69 KB of LDS, three per CU; their queues hold 8 live runs
    //  per lane, and a solve that meets more comes back here with blk_no_small set)
    bool placed = false;
    const int32_t rings_walk[] = {256, 1024, 2048}, rings_cls[] = {256, 512};
    const int32_t *rings = cls ? rings_cls : rings_walk;
    const int n_rings = cls ? 2 : 3;
    static_assert(PHI_DP_BLOCK_MAX == 2048, "the largest ring of tops");
    for (int ri = 0; ri < n_rings && !placed; ri++) {
        const int32_t ring = rings[ri];
        if (ring == 256 && c->blk_no_small) continue;
        int64_t target = (int64_t)nk * (c->n_walks + 1) / 4096;
        target = std::max<int64_t>(64, std::min<int64_t>(ring / 2, target));
        if (cls) target = c->blk_cls_target;
        if (const char *e = getenv("PHI_DP_BLOCK_STEPS")) target = std::max(1, std::min(ring, atoi(e)));   // tests: many small blocks
        c->h_blk_lo.assign(1, 0);
        bool ok = true;
        for (int32_t start = 0; nk - start > target && ok;) {
            int32_t cut = -1;
            const int32_t hi = (int32_t)std::min<int64_t>((int64_t)start + ring, nk - 1);
            for (int32_t k = (int32_t)(start + target); k <= hi && cut < 0; k++) if (cut_ok(k)) cut = k;
            for (int32_t k = (int32_t)(start + target) - 1; k > start && cut < 0; k--) if (cut_ok(k)) cut = k;
            if (cut < 0) {
                if (nk - start > ring) ok = false;             // no usable cut within a ring's length
                break;                                         // (else: the rest is one block)
            }
            c->h_blk_lo.push_back(cut);
            start = cut;
        }
        if (ok && nk - c->h_blk_lo.back() <= ring) { placed = true; c->blk_ring = ring; }
    }
    if (!placed) return PHI_OK;                                // the chain stays whole
    c->h_blk_lo.push_back(nk);
    c->n_blk = (int32_t)c->h_blk_lo.size() - 1;
    if (c->n_blk < 2) { c->n_blk = 0; return PHI_OK; }
    c->blk_max_len = 0;
    for (int32_t b = 0; b < c->n_blk; b++) c->blk_max_len = std::max(c->blk_max_len, c->h_blk_lo[(size_t)b + 1] - c->h_blk_lo[(size_t)b]);
    const size_t nbk = (size_t)c->n_blk, ls = (size_t)c->blk_ls, nrow = cls ? 65 : (size_t)(c->n_walks + 1);
    PHICHK(phi_dev_ensure(c, c->d_blk_lo, (nbk + 1) * 4));
    PHICHK(phi_dev_ensure(c, c->d_blk_ev, nbk * ls * 4));
    PHICHK(phi_dev_ensure(c, c->d_blk_S, nbk * ls * 4));
    PHICHK(phi_dev_ensure(c, c->d_blk_keys, nbk * ls * 4));
    PHICHK(phi_dev_ensure(c, c->d_blk_carry, nbk * ls * 4));
    {
        // The chain over class-lane blocks adds "no value" (NEGK) to whatever lies in the rows of class lanes a block does not
        // use, without looking: that must be a value NEGK can be added to -- not what an earlier owner of the memory left there
        // (found when a freed buffer of segment rows, holding sums of two NEGK, came back as this table: NEGK + INT32_MIN wraps
        // into a valid key).  A table allocated anew is filled with 0xC0C0C0C0 (below NEGK / 2, and NEGK plus it stays inside
        // int32); afterwards only keys and NEGK are written into it.
        const void *before = c->d_row_out.p;
        PHICHK(phi_dev_ensure(c, c->d_row_out, nbk * nrow * 64 * 4));
        if (c->d_row_out.p != before) HIPCHK(hipMemsetAsync(c->d_row_out.p, 0xC0, c->d_row_out.cap, c->stream));
    }
    PHICHK(phi_dev_ensure(c, c->d_rowend, nbk * nrow * 4));
    if (cls) {
        PHICHK(phi_dev_ensure(c, c->d_lane_walk, nbk * 64 * 4));
        PHICHK(phi_dev_ensure(c, c->d_walk_lane, nbk * ls * 4));
        PHICHK(phi_dev_ensure(c, c->d_coff, nbk * ls * 4));
        PHICHK(phi_dev_ensure(c, c->d_blk_ncls, nbk * 4));
        PHICHK(phi_dev_ensure(c, c->d_rownew, nbk * nrow * 4));
        PHICHK(phi_dev_ensure(c, c->d_rowdiag, nbk * 64 * 4));
        PHICHK(phi_dev_ensure(c, c->d_blk_bad, 64));
    }
    HIPCHK(hipMemcpyAsync(c->d_blk_lo.p, c->h_blk_lo.data(), (nbk + 1) * 4, hipMemcpyHostToDevice, c->stream));
    phi_launch_blk_ev(c->stream, c->d_blk_lo.as<int32_t>(), c->n_blk, c->d_ev_e.as<phi_ent_t>(), c->d_ev_off.as<int64_t>(), c->n_walks,
                      c->d_walk_vtx.as<int32_t>(), c->d_cvtx.as<int32_t>(), c->d_blk_ev.as<int32_t>());
    HIPCHK(hipStreamSynchronize(c->stream));                   // h_blk_lo is a member, but the launch above must have its copy
    c->dp_blocks = true;
    c->dp_cls = cls;
    if (getenv("PHI_TIMING")) fprintf(stderr, "[phi timing] solve: %d compact steps in %d blocks (rings of %d steps)%s\n", nk, c->n_blk, c->blk_ring, cls ? ", rows on class lanes" : "");
    return PHI_OK;
}

// ------------------------------------------------------------------------------------------ DP
struct DpHost {
    std::vector<int32_t> ends, bstart, ent_u, ent_h;
    std::vector<int32_t> rows, rowend, S, keys, carry;          // blocks in parallel: transfer rows, entry vectors, what the blocks report
};

// one DP launch with the given anchor weights; returns its value and the argmax path
// (wgt == nullptr: the weights are on the device already)
static int run_dp(phi_ctx *c, const std::vector<uint8_t> *wgt, DpHost &H, int64_t *value, std::vector<Seg> *segs)
{
    const int64_t n_dp = c->n_dp;
    const int64_t ne = c->n_entries;
    const int32_t nv = c->n_vtx;
    const bool events = c->dp_events;
    const int32_t n_ent = events ? c->n_k : nv;                // length of the per-step outputs
    PhiStageTimer tr("dp run");
    if (n_dp && wgt) HIPCHK(hipMemcpyAsync(c->d_a_weight.p, wgt->data(), (size_t)n_dp, hipMemcpyHostToDevice, c->stream));
    int32_t *d_dmax = c->d_dmax.as<int32_t>(), *d_bstart = c->d_bstart.as<int32_t>();
    int32_t *d_ent_src = c->d_ent.as<int32_t>(), *d_ent_h = c->d_ent.as<int32_t>() + n_ent;
    if (events) {
        // per run: prefix sums of the anchor weights -> one record per event (phi_dp_event_fill_kernel)
        {
            const int64_t nb = phi_scan_i32_num_blocks(n_dp);
            PHICHK(phi_dev_ensure(c, c->d_wpre, (size_t)(n_dp + 1) * 4));
            PHICHK(phi_dev_ensure(c, c->d_scan_blk, (size_t)nb * 4));
            PHICHK(phi_dev_ensure(c, c->d_scan_blkoff, (size_t)(nb + 1) * 8));
            phi_launch_scan_u8(c->stream, c->d_a_weight.as<uint8_t>(), n_dp, c->d_wpre.as<int32_t>(), c->d_scan_blk.as<int32_t>(), c->d_scan_blkoff.as<int64_t>());
        }
        PhiDpEventArgs A{};
        A.n_k = c->n_k; A.n_walks = c->n_walks; A.n_ev = c->n_ev;
        A.k_rec = c->d_k_rec.as<int32_t>(); A.k_in_packed = c->d_k_in.as<int32_t>();
        A.walk_off = c->d_walk_off.as<int64_t>();
        A.ev_e = c->d_ev_e.as<phi_ent_t>(); A.ev_off = c->d_ev_off.as<int64_t>();
        A.ev = c->d_ev.p;
        A.g_off = c->d_g_off.as<int64_t>(); A.g_span = c->d_g_span.as<uint8_t>(); A.a_weight = c->d_a_weight.as<uint8_t>();
        A.cost = 2 * (c->recombination / 2);                   // (c_1/2) twice, ILP_index.cpp:1276,1299
        A.dmax = d_dmax; A.bstart = d_bstart;
        A.tops = c->d_top.as<int32_t>();
        A.ent_src = d_ent_src; A.ent_h = d_ent_h;
        A.err = (uint32_t *)(c->d_scalars.as<uint64_t>() + S_ERR);
        A.q_limit = getenv("PHI_DP_QLIMIT") ? atoi(getenv("PHI_DP_QLIMIT")) : 0;      // tests: provoke the fallback
        phi_launch_dp_event_fill(c->stream, A, c->d_e_out.as<uint8_t>(), c->d_walk_vtx.as<int32_t>(), c->d_cvtx.as<int32_t>(),
                                 c->d_a_e1.as<phi_ent_t>(), c->d_wpre.as<int32_t>(), ne);
        if (tr.on) { (void)hipStreamSynchronize(c->stream); tr.lap("weights + per-run records"); }
        A.lane_stride = c->blk_ls;
        auto keep_whole_chain = [&](uint32_t kerr, uint32_t bit) -> int {
            // (both flags: block tasks that ran on clamped classes may have overflowed their queues as well)
            kerr &= ~(bit | PHI_KERR_DP_QUEUE | PHI_KERR_DP_CLASSES);
            HIPCHK(phi_copy_sync(c, c->d_scalars.as<uint64_t>() + S_ERR, &kerr, 4, hipMemcpyHostToDevice));
            if (bit == PHI_KERR_DP_CLASSES && c->blk_cls_target > 2 && !getenv("PHI_DP_BLOCK_STEPS")) {
                // a block with more than 64 classes of walks (denser variation than the block length was chosen for):
                // shorter blocks hold fewer sites; only below two steps per block does the whole chain take over
                c->blk_cls_target /= 2;
                PHICHK(dp_prepare_blocks(c, n_dp));
                return run_dp(c, wgt, H, value, segs);
            }
            if (bit == PHI_KERR_DP_QUEUE && c->blk_ring <= 256 && !c->blk_no_small) {
                // more than 8 live runs on a lane of a 256-step block task: the longer blocks have queues of 16
                c->blk_no_small = true;
                PHICHK(dp_prepare_blocks(c, n_dp));
                return run_dp(c, wgt, H, value, segs);
            }
            if (getenv("PHI_DP_STRICT")) return phi_fail(c, PHI_ERR_DEVICE, "DP blocks given up (kernel flag %u) under PHI_DP_STRICT", bit);   // tests
            c->dp_blocks = false;
            c->dp_cls = false;
            return run_dp(c, wgt, H, value, segs);
        };
        if (c->dp_blocks && c->dp_cls) {
            // more than 64 walks.  1. the walks of every block in classes (this run's weights); 2. every block from
            // every class lane: rows of its transfer matrix; 3. the chain over the blocks, on the device; 4. the blocks
            // again, on walk lanes, from their true entry vectors; 5. the two passes must agree
            const int32_t nb = c->n_blk;
            A.n_blk = nb; A.blk_ring = c->blk_ring; A.blk_max_len = c->blk_max_len; A.blk_lo = c->d_blk_lo.as<int32_t>(); A.blk_ev = c->d_blk_ev.as<int32_t>(); A.blk_S = c->d_blk_S.as<int32_t>();
            A.row_out = c->d_row_out.as<int32_t>(); A.rowend_out = c->d_rowend.as<int32_t>(); A.rownew_out = c->d_rownew.as<int32_t>();
            A.rowdiag_out = c->d_rowdiag.as<int32_t>();
            A.blk_keys_out = c->d_blk_keys.as<int32_t>(); A.blk_carry = c->d_blk_carry.as<int32_t>();
            A.lane_walk = c->d_lane_walk.as<int32_t>();
            PhiBlkClassArgs G{};
            G.n_blk = nb; G.n_walks = c->n_walks; G.lane_stride = c->blk_ls;
            G.blk_lo = A.blk_lo; G.blk_ev = A.blk_ev; G.ev_off = A.ev_off; G.walk_off = A.walk_off; G.ev = A.ev;
            G.lane_walk = c->d_lane_walk.as<int32_t>(); G.walk_lane = c->d_walk_lane.as<int32_t>(); G.coff = c->d_coff.as<int32_t>();
            G.blk_ncls = c->d_blk_ncls.as<int32_t>(); G.err = A.err;
            phi_launch_blk_classes(c->stream, G);
            uint32_t kerr = 0;
            if (tr.on) {
                HIPCHK(hipMemcpyAsync(&kerr, c->d_scalars.as<uint64_t>() + S_ERR, 4, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                tr.lap("block classes");
                if (kerr & PHI_KERR_DP_CLASSES) return keep_whole_chain(kerr, PHI_KERR_DP_CLASSES);
            }
            phi_launch_dp_block_rows(c->stream, A);
            if (tr.on) { (void)hipStreamSynchronize(c->stream); tr.lap("block rows"); }
            {
                // the chain over the blocks: one workgroup, 1.1 us per block -- or, from a few thousand blocks on, cut into segments
                // whose matrices are made in parallel, chained, and replayed in parallel (dp_events.hip; PHI_DP_CHAIN_SEGMENTS: tests)
                int32_t n_seg = nb >= 4096 ? std::min<int32_t>(64, nb / 1024) : 0;
                if (const char *e = getenv("PHI_DP_CHAIN_SEGMENTS")) n_seg = std::max(0, std::min(atoi(e), nb));
                if (n_seg >= 2) {
                    std::vector<int32_t> seg_lo((size_t)n_seg + 1);
                    for (int32_t g_ = 0; g_ <= n_seg; g_++) seg_lo[(size_t)g_] = (int32_t)((int64_t)nb * g_ / n_seg);
                    PHICHK(phi_dev_ensure(c, c->d_seg_lo, ((size_t)n_seg + 1) * 4));
                    PHICHK(phi_dev_ensure(c, c->d_seg_row, (size_t)n_seg * (size_t)(c->n_walks + 1) * (size_t)c->blk_ls * 4));
                    PHICHK(phi_dev_ensure(c, c->d_seg_S, (size_t)n_seg * (size_t)c->blk_ls * 4));
                    HIPCHK(phi_copy_sync(c, c->d_seg_lo.p, seg_lo.data(), seg_lo.size() * 4, hipMemcpyHostToDevice));
                    phi_launch_blk_chain_segments(c->stream, G, A.row_out, A.rownew_out, A.rowdiag_out, c->d_blk_S.as<int32_t>(), n_seg,
                                                  c->d_seg_lo.as<int32_t>(), c->d_seg_row.as<int32_t>(), c->d_seg_S.as<int32_t>());
                } else {
                    phi_launch_blk_chain(c->stream, G, A.row_out, A.rownew_out, A.rowdiag_out, c->d_blk_S.as<int32_t>());
                }
                HIPCHK(hipGetLastError());
            }
            if (tr.on) { (void)hipStreamSynchronize(c->stream); tr.lap("block chain"); }
            A.lane_walk = nullptr;
            phi_launch_dp_block_paths_wide(c->stream, A);
            int32_t bad[2] = {0, INT32_MAX};
            HIPCHK(hipMemcpyAsync(c->d_blk_bad.p, bad, 8, hipMemcpyHostToDevice, c->stream));
            phi_launch_blk_check(c->stream, A.blk_keys_out, A.blk_S, nb, c->blk_ls, c->n_walks, c->d_blk_bad.as<int32_t>());
            HIPCHK(hipMemcpyAsync(bad, c->d_blk_bad.p, 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(&kerr, c->d_scalars.as<uint64_t>() + S_ERR, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            if (tr.on) tr.lap("block paths + check");
            if (kerr & PHI_KERR_DP_CLASSES) return keep_whole_chain(kerr, PHI_KERR_DP_CLASSES);   // (rows and chain ran on clamped classes: void)
            if (kerr & PHI_KERR_DP_QUEUE) return keep_whole_chain(kerr, PHI_KERR_DP_QUEUE);
            if (bad[0])
                return phi_fail(c, PHI_ERR_DEVICE, "DP blocks: %d keys leaving blocks (first in block %d) differ from the chained class rows (internal error)", bad[0], bad[1]);
        } else if (c->dp_blocks) {
            // 1. every block from every entry walk (and from the walk starts inside it): rows of its transfer matrix
            const int32_t nb = c->n_blk, nwk = c->n_walks, nrow = nwk + 1;
            A.n_blk = nb; A.blk_ring = c->blk_ring; A.blk_max_len = c->blk_max_len; A.blk_lo = c->d_blk_lo.as<int32_t>(); A.blk_ev = c->d_blk_ev.as<int32_t>(); A.blk_S = c->d_blk_S.as<int32_t>();
            A.row_out = c->d_row_out.as<int32_t>(); A.rowend_out = c->d_rowend.as<int32_t>();
            A.blk_keys_out = c->d_blk_keys.as<int32_t>(); A.blk_carry = c->d_blk_carry.as<int32_t>();
            phi_launch_dp_block_rows(c->stream, A);
            H.rows.resize((size_t)nb * nrow * 64); H.rowend.resize((size_t)nb * nrow);
            HIPCHK(hipMemcpyAsync(H.rows.data(), A.row_out, H.rows.size() * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(H.rowend.data(), A.rowend_out, H.rowend.size() * 4, hipMemcpyDeviceToHost, c->stream));
            uint32_t kerr = 0;
            HIPCHK(hipMemcpyAsync(&kerr, c->d_scalars.as<uint64_t>() + S_ERR, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            if (tr.on) tr.lap("block rows");
            if (kerr & PHI_KERR_DP_QUEUE) return keep_whole_chain(kerr, PHI_KERR_DP_QUEUE);   // (more live runs than a block task's queue holds)
            // 2. chain: S_{b+1}[h'] = max(rows of walk starts, max_j S_b[j] + row_j[h'])  (max-plus, NEGK = no run)
            H.S.assign((size_t)nb * 64, PHI_DP_NEGK);
            std::vector<int64_t> cur(64, PHI_DP_NEGK), nxt(64);
            int64_t end_best = INT64_MIN;
            for (int32_t b = 0; b < nb; b++) {
                for (int h = 0; h < 64; h++) H.S[(size_t)b * 64 + h] = cur[h] > PHI_DP_NEGK / 2 ? (int32_t)cur[h] : PHI_DP_NEGK;
                std::fill(nxt.begin(), nxt.end(), (int64_t)PHI_DP_NEGK);
                for (int32_t j = 0; j <= nwk; j++) {
                    const int64_t base = j == nwk ? 0 : cur[j];
                    if (j < nwk && base <= PHI_DP_NEGK / 2) continue;
                    const int32_t *row = &H.rows[((size_t)b * nrow + j) * 64];
                    for (int32_t h2 = 0; h2 < nwk; h2++)
                        if (row[h2] > PHI_DP_NEGK / 2) nxt[h2] = std::max(nxt[h2], base + row[h2]);
                    const int32_t re = H.rowend[(size_t)b * nrow + j];
                    if (re > -(1 << 27)) end_best = std::max(end_best, base + re);
                }
                cur.swap(nxt);
            }
            // 3. the blocks again, in parallel, each from its true entry vector: everything the backtrack reads
            HIPCHK(hipMemcpyAsync(c->d_blk_S.p, H.S.data(), H.S.size() * 4, hipMemcpyHostToDevice, c->stream));
            phi_launch_dp_block_paths(c->stream, A);
            H.keys.resize((size_t)nb * 64); H.carry.resize((size_t)nb * 64);
            HIPCHK(hipMemcpyAsync(H.keys.data(), A.blk_keys_out, H.keys.size() * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(H.carry.data(), A.blk_carry, H.carry.size() * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(&kerr, c->d_scalars.as<uint64_t>() + S_ERR, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            if (kerr & PHI_KERR_DP_QUEUE) return keep_whole_chain(kerr, PHI_KERR_DP_QUEUE);
            // the two passes must agree on what leaves every block (and so on the chain as a whole)
            for (int32_t b = 0; b + 1 < nb; b++)
                for (int32_t h = 0; h < nwk; h++) {
                    const int32_t want = H.S[(size_t)(b + 1) * 64 + h], got = H.keys[(size_t)b * 64 + h];
                    // a walk that begins inside block b+1 or ended before it carries nothing that is ever read
                    if (want != got && !(want <= PHI_DP_NEGK / 2 && got <= PHI_DP_NEGK / 2))
                        return phi_fail(c, PHI_ERR_DEVICE, "DP blocks: block %d leaves key %d on walk %d, the chained matrices say %d (internal error)", b, got, h, want);
                }
            (void)end_best;
        } else {
            phi_launch_dp_events(c->stream, A);
        }
    } else {
        PHICHK(phi_dev_ensure(c, c->d_word, (size_t)c->n_entries * 8));          // (only the every-vertex kernel needs it)
        phi_launch_dp_words(c->stream, c->d_e_out.as<uint8_t>(), c->d_g_off.as<int64_t>(), c->d_g_span.as<uint8_t>(),
                            c->d_a_weight.as<uint8_t>(), ne, c->d_word.as<uint64_t>());
        PhiDpArgs A{};
        A.n_vtx = nv; A.n_walks = c->n_walks;
        A.st_rec = c->d_st_rec.as<int32_t>(); A.st_mask = c->d_st_mask.as<unsigned long long>();
        A.in_packed = c->d_in_packed.as<int32_t>();
        A.walk_off = c->d_walk_off.as<int64_t>();
        A.word = c->d_word.as<uint64_t>();
        A.g_off = c->d_g_off.as<int64_t>(); A.g_span = c->d_g_span.as<uint8_t>(); A.a_weight = c->d_a_weight.as<uint8_t>();
        A.cost = 2 * (c->recombination / 2);
        A.dmax = d_dmax; A.bstart = d_bstart;
        A.tops = c->d_top.as<int32_t>();
        A.ent_src = d_ent_src; A.ent_h = d_ent_h;
        phi_launch_dp(c->stream, A);
    }
    HIPCHK(hipGetLastError());
    // Backtracking touches a handful of entries (two per haplotype switch): read them one by one
    // instead of downloading them per walk entry; a path with very many switches (tiny R) falls
    // back to one bulk download.
    H.ends.resize(c->n_walks);
    phi_launch_gather_i32(c->stream, d_dmax, c->d_walk_last.as<phi_ent_t>(), c->n_walks, c->d_list3.as<int32_t>());
    HIPCHK(hipMemcpyAsync(H.ends.data(), c->d_list3.p, (size_t)c->n_walks * 4, hipMemcpyDeviceToHost, c->stream));
    uint32_t kerr = 0;
    if (events) HIPCHK(hipMemcpyAsync(&kerr, c->d_scalars.as<uint64_t>() + S_ERR, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    tr.lap("kernel");
    if (kerr & PHI_KERR_DP_QUEUE) {
        // a lane of the four-wave event kernel had more live runs than its queue holds: this graph takes
        // the every-vertex kernel from now on
        kerr &= ~PHI_KERR_DP_QUEUE;
        HIPCHK(phi_copy_sync(c, c->d_scalars.as<uint64_t>() + S_ERR, &kerr, 4, hipMemcpyHostToDevice));
        if (!c->dp_dense_ready) return phi_fail(c, PHI_ERR_DEVICE, "event DP queue overflow without a dense fallback (internal error)");
        c->dp_events = false;
        return run_dp(c, wgt, H, value, segs);
    }
    bool bulk = false;
    auto fetch_bulk = [&]() -> int {
        H.bstart.resize(ne); H.ent_u.resize(n_ent); H.ent_h.resize(n_ent);
        HIPCHK(hipMemcpyAsync(H.bstart.data(), d_bstart, (size_t)ne * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(H.ent_u.data(), d_ent_src, (size_t)n_ent * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(H.ent_h.data(), d_ent_h, (size_t)n_ent * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        bulk = true;
        return PHI_OK;
    };

    // end at (last(h), h) for the best h; ties go to the lowest walk id
    int32_t bh = -1;
    int64_t best = INT64_MIN;
    for (int32_t h = 0; h < c->n_walks; h++) {
        const int32_t d = H.ends[h];
        if (d > -(1 << 28) && d > best) { best = d; bh = h; }
    }
    if (bh < 0) return phi_fail(c, PHI_ERR_DEVICE, "DP found no s->e path (internal error)");
    *value = best;
    segs->clear();
    int32_t h = bh;
    int64_t e = c->h_walk_off[h + 1] - 1;
    for (int64_t guard = 0; guard <= (int64_t)nv; guard++) {
        if (!bulk && guard == 64) PHICHK(fetch_bulk());
        int32_t bs = 0;
        if (bulk) bs = H.bstart[e];
        else HIPCHK(phi_copy_sync(c, &bs, d_bstart + e, 4, hipMemcpyDeviceToHost));
        if (bs < 0 && c->dp_blocks && events) {
            // the run crossed into its block: follow it back through the blocks it was carried over
            int32_t vq = 0;
            PHICHK(walk_vtx_at(c, e, &vq));
            const int32_t kq = c->h_cstep[c->h_topo_rank[vq]];
            int32_t b = (int32_t)(std::upper_bound(c->h_blk_lo.begin(), c->h_blk_lo.end(), kq) - c->h_blk_lo.begin()) - 1;
            if (c->dp_cls) {
                int32_t out[2] = {-1, -1};
                phi_launch_carry_resolve(c->stream, c->d_blk_carry.as<int32_t>(), c->blk_ls, b, h, c->d_blk_bad.as<int32_t>() + 2);
                HIPCHK(hipMemcpyAsync(out, c->d_blk_bad.as<int32_t>() + 2, 8, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                bs = out[1];
            } else {
                while (bs < 0 && --b >= 0) bs = H.carry[(size_t)b * 64 + h];
            }
            if (bs < 0) return phi_fail(c, PHI_ERR_DEVICE, "DP backtrack: a carried run has no beginning (internal error)");
        }
        const int64_t es = c->h_walk_off[h] + bs;
        if (es < c->h_walk_off[h] || es > e) return phi_fail(c, PHI_ERR_DEVICE, "DP backtrack left the walk (internal error)");
        segs->push_back(Seg{h, es, e});
        if (es == c->h_walk_off[h]) break;                     // reached s_{first(h),h}
        int32_t v = 0;
        PHICHK(walk_vtx_at(c, es, &v));
        const int32_t step = events ? c->h_cstep[c->h_topo_rank[v]] : c->h_topo_rank[v];
        if (step < 0) return phi_fail(c, PHI_ERR_DEVICE, "DP backtrack: run begins on a vertex without events (internal error)");
        int32_t src, h2;
        if (bulk) { src = H.ent_u[step]; h2 = H.ent_h[step]; }
        else {
            HIPCHK(phi_copy_sync(c, &src, d_ent_src + step, 4, hipMemcpyDeviceToHost));
            HIPCHK(phi_copy_sync(c, &h2, d_ent_h + step, 4, hipMemcpyDeviceToHost));
        }
        if (events && src >= 0) src = c->h_kstep[src];          // compact step -> topological step
        const int32_t u = src >= 0 ? c->h_topo[src] : -1;
        if (u < 0 || h2 < 0) return phi_fail(c, PHI_ERR_DEVICE, "DP backtrack hit a vertex without an entry (internal error)");
        // entry of walk h2 on vertex u: topological ranks increase along a walk
        int64_t e2 = -1;
        {
            int64_t lo = c->h_walk_off[h2], hi = c->h_walk_off[h2 + 1];
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                int32_t vm = 0;
                PHICHK(walk_vtx_at(c, mid, &vm));
                if (c->h_topo_rank[vm] < src) lo = mid + 1; else hi = mid;
            }
            if (lo < c->h_walk_off[h2 + 1]) {
                int32_t vl = 0;
                PHICHK(walk_vtx_at(c, lo, &vl));
                if (vl == u) e2 = lo;
            }
        }
        if (e2 < 0) return phi_fail(c, PHI_ERR_DEVICE, "DP backtrack: walk %d is not on vertex %d (internal error)", h2, u);
        h = h2; e = e2;
    }
    std::reverse(segs->begin(), segs->end());
    tr.lap("backtrack");
    return PHI_OK;
}

// dp anchors traversed by a path: calls f(anchor index)
template <class F> static void for_covered(const phi_ctx *c, const std::vector<Seg> &segs, F f)
{
    const PhiAnchorSpan &A = c->h_dp;                         // sorted by e1
    for (const Seg &s : segs) {
        auto lo = std::lower_bound(A.begin(), A.end(), s.es, [](const PhiAnchorHost &a, int64_t e) { return (int64_t)a.e1 < e; });
        for (auto it = lo; it != A.end() && (int64_t)it->e1 <= s.ee; ++it)
            if ((int64_t)it->e0 >= s.es) f((int64_t)(it - A.begin()));
    }
}

// ------------------------------------------------------------------------------------------ solve
struct Node {
    std::vector<std::pair<uint32_t, int32_t>> assign;          // minimiser slot -> chosen cluster
    int64_t ub = INT64_MAX;                                    // bound proven for the parent (holds for the child)
};

int phi_solve_impl(phi_ctx *c)
{
    c->solved = false;
    if (c->dp_alloc_future.valid()) PHICHK(c->dp_alloc_future.get());     // (a phi_set_graph that failed midway left it behind)
    PHICHK(phi_sync_check(c));
    PhiStageTimer tm("solve");
    uint64_t sc[S_N];
    HIPCHK(phi_copy_sync(c, sc, c->d_scalars.p, sizeof sc, hipMemcpyDeviceToHost));
    uint64_t n_distinct = 0;
    PHICHK(phi_spectrum_count(c, &n_distinct));                   // (enters the logged novel read hashes into the set: the read set's one de-duplication)
    tm.lap("read spectrum (set from the log)");
    const int64_t spectrum = c->spectrum_override >= 0 ? c->spectrum_override : (int64_t)n_distinct;
    const int64_t n_rec = c->n_rec;
    const int32_t nw = c->n_walks;

    // ---- 1. anchors = walk minimisers whose hash is in the read spectrum (:495-526)
    int64_t n_matched = 0;
    PHICHK(phi_dev_ensure(c, c->d_flags, (size_t)std::max<int64_t>(n_rec, 1)));
    phi_launch_match_flags(c->stream, c->d_rec_slot.as<uint32_t>(), n_rec, c->d_u_uid.as<uint32_t>(),
                           c->d_hit.as<uint8_t>(), c->d_flags.as<uint8_t>());
    PHICHK(phi_compact(c, c->d_flags.as<uint8_t>(), n_rec, c->d_m_rec, &n_matched));

    tm.lap("match + compact");
    // ---- 2. shared-anchor filter (:670-743)
    PhiFilterArgs F{};
    // (over the CLASS records of contexts.hip: a record stands for cls_mult walk entries; its vertex list is read
    //  at the class representative's entries)
    F.rec_slot = c->d_rec_slot.as<uint32_t>(); F.rec_e0 = c->d_rec_e0.as<phi_ent_t>(); F.rec_e1 = c->d_rec_e1.as<phi_ent_t>();
    F.walk_vtx = c->d_walk_vtx.as<int32_t>();
    F.rec_cls = c->d_rec_cls.as<int32_t>(); F.cls_mult = c->d_cls_mult.as<int32_t>();
    F.m_rec = c->d_m_rec.as<int32_t>();
    const uint64_t g_cap = pow2_at_least(std::max<uint64_t>(1024, 2 * (uint64_t)n_matched));
    PHICHK(phi_dev_ensure(c, c->d_g_keys, g_cap * 8));
    PHICHK(phi_dev_ensure(c, c->d_g_rep, g_cap * 4));
    PHICHK(phi_dev_ensure(c, c->d_g_cnt, g_cap * 4));
    PHICHK(phi_dev_ensure(c, c->d_m_group, (size_t)std::max<int64_t>(n_matched, 1) * 4));
    const size_t n_slot_ids = (size_t)std::max<int64_t>(c->n_unique, 1);
    PHICHK(phi_dev_ensure(c, c->d_slot_maxcnt, n_slot_ids * 4));
    PHICHK(phi_dev_ensure(c, c->d_slot_multi, n_slot_ids));
    F.g_keys = c->d_g_keys.as<uint64_t>(); F.g_rep = c->d_g_rep.as<int32_t>(); F.g_cnt = c->d_g_cnt.as<uint32_t>();
    F.g_mask = g_cap - 1;
    F.m_group = c->d_m_group.as<int32_t>();
    F.u_uid = c->d_u_uid.as<uint32_t>();
    F.slot_maxcnt = c->d_slot_maxcnt.as<uint32_t>(); F.slot_multi = c->d_slot_multi.as<uint8_t>();
    F.limit = c->threshold * (float)(uint32_t)nw;              // threshold * num_walks, float (:698)
    F.counters = (unsigned long long *)scalar(c, S_FILTERED);
    F.err = (uint32_t *)scalar(c, S_ERR);
    for (int attempt = 0;; attempt++) {
        F.seed = 0x243F6A8885A308D3ull + 0x9E3779B97F4A7C15ull * (uint64_t)attempt;
        phi_launch_fill_u64(c->stream, F.g_keys, (int64_t)g_cap, PHI_EMPTY_KEY);
        phi_launch_fill_u32(c->stream, (uint32_t *)F.g_rep, (int64_t)g_cap, 0xFFFFFFFFu);
        HIPCHK(hipMemsetAsync(F.g_cnt, 0, g_cap * 4, c->stream));
        phi_launch_group_insert(c->stream, F, n_matched);
        phi_launch_group_count(c->stream, F, n_matched);
        HIPCHK(hipStreamSynchronize(c->stream));
        uint32_t err = 0;
        HIPCHK(phi_copy_sync(c, &err, scalar(c, S_ERR), 4, hipMemcpyDeviceToHost));
        if (err & PHI_KERR_TABLE_FULL) return phi_fail(c, PHI_ERR_OVERFLOW, "anchor group table overflow");
        if (!(err & PHI_KERR_FP_COLLISION)) break;
        if (attempt == 7) return phi_fail(c, PHI_ERR_DEVICE, "anchor fingerprints collide under 8 seeds (internal error)");
        HIPCHK(phi_memset_sync(c, scalar(c, S_ERR), 0, 4));
    }
    HIPCHK(hipMemsetAsync(F.slot_maxcnt, 0, n_slot_ids * 4, c->stream));
    HIPCHK(hipMemsetAsync(F.slot_multi, 0, n_slot_ids, c->stream));
    HIPCHK(hipMemsetAsync(scalar(c, S_FILTERED), 0, 16, c->stream));
    phi_launch_group_max(c->stream, F, n_matched);
    phi_launch_slot_count(c->stream, F, c->n_unique);
    PHICHK(phi_dev_ensure(c, c->d_flags, (size_t)std::max<int64_t>(std::max(n_matched, n_rec), 1)));
    PHICHK(phi_dev_ensure(c, c->d_flags2, (size_t)std::max<int64_t>(n_matched, 1)));
    phi_launch_kept_flags(c->stream, F, n_matched, c->d_flags.as<uint8_t>(), c->d_flags2.as<uint8_t>());
    int64_t n_kept_rec = 0, n_kept = 0;
    bool anchors_prepped = false;
    PHICHK(phi_compact(c, c->d_flags.as<uint8_t>(), n_matched, c->d_list, &n_kept_rec));
    tm.lap("filter kernels");
    // The kept class records are expanded into the anchors of the model: every walk entry of a record's class
    // carries one, (minimiser id, first entry, last entry), in entry order = position order along the walks.
    // They go to the host as triples; their hashes stay behind (phi_kept_anchors derives them from the ids).
    {
        const int64_t nr = std::max<int64_t>(n_rec, 1);
        PHICHK(phi_dev_ensure(c, c->d_flags2, (size_t)std::max<int64_t>(nr, n_matched)));
        HIPCHK(hipMemsetAsync(c->d_flags2.p, 0, (size_t)nr, c->stream));
        phi_launch_mark_list(c->stream, c->d_list.as<int32_t>(), c->d_m_rec.as<int32_t>(), n_kept_rec, c->d_flags2.as<uint8_t>());
        PHICHK(phi_dev_ensure(c, c->d_list3, (size_t)std::max<int64_t>(c->n_cls, (int64_t)nw) * 4));
        phi_launch_class_sel_count(c->stream, c->d_flags2.as<uint8_t>(), c->d_cls_rec_off.as<int32_t>(), c->n_cls, c->d_list3.as<int32_t>());
        PhiExpandArgs X{};
        X.ent_cls = c->d_ent_cls.as<int32_t>(); X.e_lo = 0; X.e_hi = c->n_entries;
        X.cls_rec_off = c->d_cls_rec_off.as<int32_t>(); X.cls_rep = c->d_cls_rep.as<phi_ent_t>();
        X.sel = c->d_flags2.as<uint8_t>(); X.sel_cnt = c->d_list3.as<int32_t>();
        X.rec_slot = c->d_rec_slot.as<uint32_t>(); X.u_uid = c->d_u_uid.as<uint32_t>();
        X.rec_e0 = c->d_rec_e0.as<phi_ent_t>(); X.rec_e1 = c->d_rec_e1.as<phi_ent_t>();
        const int64_t nb = phi_expand_num_blocks(c->n_entries);
        PHICHK(phi_dev_ensure(c, c->d_blk_cnt, (size_t)nb * 4));
        PHICHK(phi_dev_ensure(c, c->d_blk_off, (size_t)(nb + 1) * 8));
        X.block_cnt = c->d_blk_cnt.as<int32_t>(); X.block_off = c->d_blk_off.as<int64_t>();
        phi_launch_expand_count(c->stream, X);
        PHICHK(phi_scan_counts_wide(c, c->d_blk_cnt.as<int32_t>(), nb, c->d_blk_off.as<int64_t>()));
        HIPCHK(hipMemcpyAsync(&n_kept, c->d_blk_off.as<int64_t>() + nb, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (n_kept >= (int64_t)1 << 31) return phi_fail(c, PHI_ERR_UNSUPPORTED, "more than 2^31 anchors in the model");
        c->h_kept = PhiAnchorSpan{}; c->h_dp = PhiAnchorSpan{}; c->anchors_host = false;
        c->n_kept = n_kept; c->n_dp = 0;
        c->h_kept_hash.clear();
        PHICHK(phi_dev_ensure(c, c->d_anchors, (size_t)std::max<int64_t>(n_kept, 1) * 12));
        static_assert(sizeof(PhiAnchorHost) == 12, "PhiAnchorHost is the device triple");
        // (the DP's per-anchor arrays and the counters of the anchor checks: the packed expansion fills them on its way)
        PHICHK(phi_dev_ensure(c, c->d_a_e1, (size_t)std::max<int64_t>(n_kept, 1) * 4));
        PHICHK(phi_dev_ensure(c, c->d_g_span, (size_t)std::max<int64_t>(n_kept, 1)));
        PHICHK(phi_dev_ensure(c, c->d_ctr, (size_t)(nw + 8) * 8));
        HIPCHK(hipMemsetAsync(c->d_ctr.p, 0, (size_t)(nw + 8) * 8, c->stream));
        if (n_kept && getenv("PHI_EXPAND_GENERIC")) {               // (tests: the generic expansion, record by record)
            X.out_tri = c->d_anchors.as<uint32_t>();
            phi_launch_expand_write(c->stream, X, 1);
        } else if (n_kept) {
            // the selected records packed per class, then the anchors from those alone (contexts.hip)
            PHICHK(phi_dev_ensure(c, c->d_sel_off, (size_t)(c->n_cls + 1) * 4));
            PHICHK(phi_dev_ensure(c, c->d_sel_tri, (size_t)std::max<int64_t>(n_kept_rec, 1) * 12));
            const int64_t nsb = phi_scan_i32_num_blocks(c->n_cls);
            PHICHK(phi_dev_ensure(c, c->d_scan_blk, (size_t)nsb * 4));
            PHICHK(phi_dev_ensure(c, c->d_scan_blkoff, (size_t)(nsb + 1) * 8));
            phi_launch_scan_i32(c->stream, c->d_list3.as<int32_t>(), c->n_cls, c->d_sel_off.as<int32_t>(), c->d_scan_blk.as<int32_t>(), c->d_scan_blkoff.as<int64_t>());
            phi_launch_class_sel_tri(c->stream, c->d_flags2.as<uint8_t>(), c->d_cls_rec_off.as<int32_t>(), c->n_cls, c->d_sel_off.as<int32_t>(), c->d_cls_rep.as<phi_ent_t>(),
                                     c->d_rec_slot.as<uint32_t>(), c->d_u_uid.as<uint32_t>(), c->d_rec_e0.as<phi_ent_t>(), c->d_rec_e1.as<phi_ent_t>(), c->d_sel_tri.as<int32_t>());
            unsigned long long *d_ctr = c->d_ctr.as<unsigned long long>();
            phi_launch_expand_tri(c->stream, c->d_ent_cls.as<int32_t>(), 0, c->n_entries, c->d_sel_off.as<int32_t>(), c->d_sel_tri.as<int32_t>(),
                                  c->d_blk_off.as<int64_t>(), c->d_anchors.as<uint32_t>(), c->d_a_e1.as<phi_ent_t>(), c->d_g_span.as<uint8_t>(),
                                  c->d_walk_off.as<int64_t>(), nw, d_ctr + 8, d_ctr, n_kept);
            anchors_prepped = true;
        }
    }
    // The DP's per-anchor arrays (last entry, span), the anchors per walk and the checks on them, on the device.
    // A large model whose anchors all span an edge (vertices shorter than k: every graph chopped to 30 bp) stays
    // there: the host copy (6 GB at 5 * 10^8 anchors) is fetched only if the branch and bound proper needs it.
    const uint32_t *d_tri = c->d_anchors.as<uint32_t>();
    bool dev = false;
    {
        unsigned long long *d_ctr = c->d_ctr.as<unsigned long long>();
        if (!anchors_prepped)
            phi_launch_anchor_prep(c->stream, d_tri, n_kept, c->d_walk_off.as<int64_t>(), nw, c->d_a_e1.as<phi_ent_t>(), c->d_g_span.as<uint8_t>(), d_ctr + 8, d_ctr);
        std::vector<unsigned long long> hc((size_t)nw + 8);
        HIPCHK(hipMemcpyAsync(hc.data(), d_ctr, hc.size() * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (hc[1] & 1) return phi_fail(c, PHI_ERR_DEVICE, "dp anchors not sorted by last entry (internal error)");
        if (hc[1] & 2)
            return phi_fail(c, c->k > PHI_RCAP ? PHI_ERR_UNSUPPORTED : PHI_ERR_DEVICE, "an anchor spans %d edges or more%s", PHI_RCAP,
                            c->k > PHI_RCAP ? ": with k > 32 the graph's vertices must be long enough for a k-mer to cover at most 32 of them" : " (internal error)");
        c->h_n_anchors.assign(nw, 0);
        for (int32_t h = 0; h < nw; h++) c->h_n_anchors[h] = (int64_t)hc[(size_t)h + 8];
        int64_t dev_min = (int64_t)1 << 16;                    // below this the host loops are as fast as the extra launches
        if (const char *e = getenv("PHI_SOLVE_DEVICE")) dev_min = atoi(e) ? 0 : INT64_MAX;     // tests: force either way
        dev = hc[0] == 0 && n_kept >= dev_min && n_kept > 0;
    }
    HIPCHK(phi_copy_sync(c, sc, c->d_scalars.p, sizeof sc, hipMemcpyDeviceToHost));
    const int64_t filtered = (int64_t)sc[S_FILTERED], in_model = (int64_t)sc[S_INMODEL];
    if (!dev) {
        PHICHK(phi_pin_ensure(c, (size_t)std::max<int64_t>(n_kept, 1) * sizeof(PhiAnchorHost)));
        c->h_kept = PhiAnchorSpan{static_cast<PhiAnchorHost *>(c->h_pin), n_kept};
        if (n_kept) HIPCHK(hipMemcpyAsync(c->h_kept.p, d_tri, (size_t)n_kept * 12, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->anchors_host = true;
    }

    tm.lap(dev ? "anchor arrays (device)" : "anchors D2H");
    // dp anchors (span >= 1 edge; single-vertex anchors are ignored, :795/:846), anchors per walk and the
    // DP's per-anchor arrays: host threads over chunks of the kept list, order preserved
    std::vector<phi_ent_t> a_e1;
    std::vector<uint8_t> a_span;
    std::vector<int16_t> dp_walk;                              // walk of every dp anchor
    if (dev) c->n_dp = n_kept;
    else {
        c->h_n_anchors.assign(nw, 0);
        const int64_t chunk = (int64_t)1 << 16;
        const int64_t n_chunks = (n_kept + chunk - 1) / chunk;
        std::vector<int64_t> dp_cnt(n_chunks + 1, 0);
        std::vector<std::vector<int32_t>> walk_cnt(phi_host_threads(), std::vector<int32_t>(nw, 0));
        phi_parallel_chunks(n_kept, chunk, [&](int64_t lo, int64_t hi, int worker) {
            int64_t n = 0;
            int32_t hw = phi_entry_walk(c, c->h_kept[lo].e0);
            for (int64_t i = lo; i < hi; i++) {
                const PhiAnchorHost &k = c->h_kept[i];
                n += k.e1 > k.e0;
                while (k.e0 >= c->h_walk_off[hw + 1]) hw++;      // kept anchors come in walk order
                walk_cnt[worker][hw]++;
            }
            dp_cnt[lo / chunk + 1] = n;
        });
        for (int64_t i = 0; i < n_chunks; i++) dp_cnt[i + 1] += dp_cnt[i];
        for (const auto &wc : walk_cnt) for (int32_t h = 0; h < nw; h++) c->h_n_anchors[h] += wc[h];
        const int64_t n_dp0 = dp_cnt[n_chunks];
        // usually every kept anchor spans an edge (vertices shorter than k): the dp list is the kept list
        const bool same = n_dp0 == n_kept;
        if (same) { c->h_dp_own.clear(); c->h_dp = c->h_kept; }
        else { c->h_dp_own.resize(n_dp0); c->h_dp = PhiAnchorSpan{c->h_dp_own.data(), n_dp0}; }
        dp_walk.resize(n_dp0);
        a_e1.resize(n_dp0);
        a_span.resize(n_dp0);
        PhiHostError herr;
        phi_parallel_chunks(n_kept, chunk, [&](int64_t lo, int64_t hi, int) {
            int64_t o = dp_cnt[lo / chunk];
            int32_t hw = phi_entry_walk(c, c->h_kept[lo].e0);
            for (int64_t i = lo; i < hi; i++) {
                const PhiAnchorHost &k = c->h_kept[i];
                while (k.e0 >= c->h_walk_off[hw + 1]) hw++;
                if (k.e1 <= k.e0) continue;
                dp_walk[o] = (int16_t)hw;
                if (k.e1 - k.e0 >= (uint32_t)PHI_RCAP) { herr.set(c->k > PHI_RCAP ? PHI_ERR_UNSUPPORTED : PHI_ERR_DEVICE, "an anchor spans %u edges (at most %d are supported)", k.e1 - k.e0, PHI_RCAP - 1); return; }
                if (!same) c->h_dp[o] = k;
                a_e1[o] = k.e1;
                a_span[o] = (uint8_t)(k.e1 - k.e0);
                o++;
            }
        });
        if (herr.failed()) return phi_fail(c, herr.code, "%s", herr.msg.c_str());
        phi_parallel_chunks(n_dp0, chunk, [&](int64_t lo, int64_t hi, int) {
            for (int64_t i = std::max<int64_t>(lo, 1); i < hi; i++)
                if (a_e1[i] < a_e1[i - 1]) { herr.set(PHI_ERR_DEVICE, "dp anchors not sorted by last entry (internal error)"); return; }
        });
        if (herr.failed()) return phi_fail(c, herr.code, "%s", herr.msg.c_str());
        c->n_dp = (int64_t)c->h_dp.size();
    }
    const int64_t n_dp = c->n_dp;
    // DP scores are int32 with -2^28 as "no state": a path scores at most one per anchor
    if (n_dp >= ((int64_t)1 << 27) && !dev) {
        // a path is on one walk at every vertex, so it scores at most sum over vertices of the most anchors that
        // end there on one walk: that sum, not the number of anchors, has to stay inside the DP's score range
        std::vector<int32_t> vmax((size_t)c->n_vtx, 0);
        PHICHK(ensure_host_walks(c));
        for (int64_t i = 0; i < n_dp;) {
            int64_t j = i;
            while (j < n_dp && a_e1[j] == a_e1[i]) j++;
            int32_t &m = vmax[c->h_walk_vtx[a_e1[i]]];
            m = std::max<int32_t>(m, (int32_t)std::min<int64_t>(j - i, INT32_MAX));
            i = j;
        }
        int64_t bound = 0;
        for (int32_t m : vmax) bound += m;
        if (bound >= ((int64_t)1 << 27)) return phi_fail(c, PHI_ERR_UNSUPPORTED, "a path could score %lld >= 2^27 anchors", (long long)bound);
    }

    if (tm.on) fprintf(stderr, "[phi timing] solve: class records %lld, matched %lld, kept %lld -> anchors kept %lld, dp %lld\n", (long long)n_rec, (long long)n_matched, (long long)n_kept_rec, (long long)n_kept, (long long)n_dp);
    tm.lap("filter (GPU) + anchors D2H");
    // ---- 3. DP inputs
    {
        PHICHK(phi_dev_ensure(c, c->d_a_weight, (size_t)std::max<int64_t>(n_dp, 1)));
        PHICHK(phi_dev_ensure(c, c->d_g_off, (size_t)(c->n_entries + 1) * 8));
        if (n_dp && !dev) {
            // (the dp list may be shorter than the kept list the device arrays were made from)
            HIPCHK(hipMemcpyAsync(c->d_a_e1.p, a_e1.data(), (size_t)n_dp * 4, hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipMemcpyAsync(c->d_g_span.p, a_span.data(), (size_t)n_dp, hipMemcpyHostToDevice, c->stream));
        }
        phi_launch_entry_csr(c->stream, c->d_a_e1.as<phi_ent_t>(), n_dp, c->n_entries, c->d_g_off.as<int64_t>());
        if (dev && n_dp >= ((int64_t)1 << 27)) {
            // the score range, as above: sum over vertices of the most anchors that end there on one walk
            PHICHK(phi_dev_ensure(c, c->d_vmax, (size_t)c->n_vtx * 4));
            HIPCHK(hipMemsetAsync(c->d_vmax.p, 0, (size_t)c->n_vtx * 4, c->stream));
            HIPCHK(hipMemsetAsync(c->d_ctr.p, 0, 8, c->stream));
            phi_launch_vertex_most(c->stream, c->d_g_off.as<int64_t>(), c->n_entries, c->d_walk_vtx.as<int32_t>(), c->d_vmax.as<int32_t>());
            phi_launch_sum_i32(c->stream, c->d_vmax.as<int32_t>(), c->n_vtx, c->d_ctr.as<unsigned long long>());
            unsigned long long bound = 0;
            HIPCHK(hipMemcpyAsync(&bound, c->d_ctr.p, 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            if (bound >= (1ull << 27)) return phi_fail(c, PHI_ERR_UNSUPPORTED, "a path could score %llu >= 2^27 anchors", bound);
        }
        HIPCHK(hipStreamSynchronize(c->stream));
        {
            std::vector<phi_ent_t> last(nw);
            for (int32_t h = 0; h < nw; h++) last[h] = (phi_ent_t)(c->h_walk_off[h + 1] - 1);
            PHICHK(phi_dev_ensure(c, c->d_walk_last, (size_t)nw * 4));
            PHICHK(phi_dev_ensure(c, c->d_list3, (size_t)nw * 4));
            HIPCHK(phi_copy_sync(c, c->d_walk_last.p, last.data(), (size_t)nw * 4, hipMemcpyHostToDevice));
        }
        PHICHK(phi_dev_ensure(c, c->d_dmax, (size_t)c->n_entries * 4));
        PHICHK(phi_dev_ensure(c, c->d_bstart, (size_t)c->n_entries * 4));
        if (c->dp_events) {
            const int64_t ne1 = c->n_entries + 1;
            PHICHK(phi_dev_ensure(c, c->d_off_end, (size_t)(ne1 + 2) * 4));      // (+2: also the scratch of dp_prepare_blocks)
            PHICHK(phi_dev_ensure(c, c->d_off_start, (size_t)(ne1 + 2) * 4));
            const int64_t nb = phi_scan_i32_num_blocks(c->n_entries);
            PHICHK(phi_dev_ensure(c, c->d_scan_blk, (size_t)nb * 4));
            PHICHK(phi_dev_ensure(c, c->d_scan_blkoff, (size_t)(nb + 1) * 8));
            PHICHK(phi_dev_ensure(c, c->d_ev, (size_t)std::max<int64_t>(c->n_ev, 1) * 48));
        }
        PHICHK(phi_dev_ensure(c, c->d_top, (size_t)c->n_vtx * 5 * 4));
        PHICHK(phi_dev_ensure(c, c->d_ent, (size_t)c->n_vtx * 2 * 4));
    }

    c->blk_no_small = false;
    c->blk_cls_target = 16;
    PHICHK(dp_prepare_blocks(c, n_dp));
    tm.lap("DP inputs");
    // ---- 4. exact solve
    const int64_t cost = 2 * (int64_t)(c->recombination / 2);
    // dp anchors of every minimiser (CSR over the dense minimiser ids), ascending anchor index
    const int64_t n_ids = c->n_unique;
    std::vector<int32_t> sa_off, sa_idx;                       // (device mode: on the device only, until the host needs them)
    if (!dev) { sa_off.assign((size_t)n_ids + 1, 0); sa_idx.resize((size_t)std::max<int64_t>(n_dp, 1)); }
    const bool dp_is_kept = dev || (c->h_dp.p == c->h_kept.p && c->h_dp.size() == c->h_kept.size());
    // (device mode does without the map -- repeats walk by walk, weights from a flag per minimiser -- until the branch and
    //  bound proper asks for the host copies: at chromosome scale building it is 0.16 of the solve's 0.95 s)
    auto build_csr = [&](bool to_host_too) -> int {
        // the dp list is the kept list, whose triples are on the device: count / scan / scatter /
        // sort there (1-2 ms for 10^7 anchors; the host loop below takes 4 ms per million)
        const uint32_t *tri = d_tri;
        PHICHK(phi_dev_ensure(c, c->d_sa_cnt, (size_t)(n_ids + 1) * 4));
        PHICHK(phi_dev_ensure(c, c->d_sa_cur, (size_t)(n_ids + 1) * 4));
        PHICHK(phi_dev_ensure(c, c->d_sa_off, (size_t)(n_ids + 2) * 4));
        PHICHK(phi_dev_ensure(c, c->d_sa_idx, (size_t)n_dp * 4));
        const int64_t nb = phi_scan_i32_num_blocks(n_ids);
        PHICHK(phi_dev_ensure(c, c->d_scan_blk, (size_t)nb * 4));
        PHICHK(phi_dev_ensure(c, c->d_scan_blkoff, (size_t)(nb + 1) * 8));
        HIPCHK(hipMemsetAsync(c->d_sa_cnt.p, 0, (size_t)(n_ids + 1) * 4, c->stream));
        HIPCHK(hipMemsetAsync(c->d_sa_cur.p, 0, (size_t)(n_ids + 1) * 4, c->stream));
        phi_launch_csr_count(c->stream, tri, n_dp, n_ids, c->d_sa_cnt.as<int32_t>(), (uint32_t *)scalar(c, S_ERR));
        phi_launch_scan_i32(c->stream, c->d_sa_cnt.as<int32_t>(), n_ids, c->d_sa_off.as<int32_t>(), c->d_scan_blk.as<int32_t>(),
                            c->d_scan_blkoff.as<int64_t>());
        phi_launch_csr_scatter(c->stream, tri, n_dp, n_ids, c->d_sa_off.as<int32_t>(), c->d_sa_cur.as<int32_t>(), c->d_sa_idx.as<int32_t>());
        phi_launch_csr_sort(c->stream, c->d_sa_off.as<int32_t>(), n_ids, c->d_sa_idx.as<int32_t>());
        int32_t total = 0;
        if (to_host_too) {
            sa_off.resize((size_t)n_ids + 1); sa_idx.resize((size_t)std::max<int64_t>(n_dp, 1));
            HIPCHK(hipMemcpyAsync(sa_off.data(), c->d_sa_off.p, (size_t)(n_ids + 1) * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(sa_idx.data(), c->d_sa_idx.p, (size_t)n_dp * 4, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(hipMemcpyAsync(&total, c->d_sa_off.as<int32_t>() + n_ids, 4, hipMemcpyDeviceToHost, c->stream));
        uint32_t kerr = 0;
        HIPCHK(hipMemcpyAsync(&kerr, scalar(c, S_ERR), 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if ((kerr & PHI_KERR_CSR_ID) || total != (int32_t)n_dp)
            return phi_fail(c, PHI_ERR_DEVICE, "minimiser id out of range (internal error)");
        return PHI_OK;
    };
    if (dev) {
        // (nothing yet)
    } else if (dp_is_kept && n_dp > 0) {
        PHICHK(build_csr(true));
    } else {
        // (tried on host threads with atomic counters and per-list sorts: 7.5 ms against 4.3 ms for this loop)
        for (const PhiAnchorHost &a : c->h_dp) {
            if ((int64_t)a.slot >= n_ids) return phi_fail(c, PHI_ERR_DEVICE, "minimiser id out of range (internal error)");
            sa_off[a.slot + 1]++;
        }
        for (int64_t i = 0; i < n_ids; i++) sa_off[i + 1] += sa_off[i];
        std::vector<int32_t> cur(sa_off.begin(), sa_off.end() - 1);
        for (int64_t i = 0; i < n_dp; i++) sa_idx[cur[c->h_dp[i].slot]++] = (int32_t)i;
    }
    struct Span {
        const int32_t *p; int32_t n;
        const int32_t *begin() const { return p; }
        const int32_t *end() const { return p + n; }
        int32_t operator[](int32_t i) const { return p[i]; }
    };
    auto anchors_of = [&](uint32_t s) { return Span{sa_idx.data() + sa_off[s], sa_off[s + 1] - sa_off[s]}; };
    // clusters of pairwise mutually exclusive anchors of one minimiser: anchors on different walks
    // whose topological-rank intervals overlap cannot both be traversed (a path holds one
    // haplotype label per vertex).  Anything else becomes a singleton cluster.
    auto clusters_of = [&](uint32_t slot) {
        const Span idx = anchors_of(slot);
        struct Iv { int32_t lo, hi, h, a; };
        std::vector<Iv> iv;
        for (int32_t a : idx) {
            const PhiAnchorHost &A = c->h_dp[a];
            iv.push_back(Iv{c->h_topo_rank[c->h_walk_vtx[A.e0]], c->h_topo_rank[c->h_walk_vtx[A.e1]], phi_entry_walk(c, A.e0), a});
        }
        std::sort(iv.begin(), iv.end(), [](const Iv &x, const Iv &y) { return x.lo != y.lo ? x.lo < y.lo : x.a < y.a; });
        std::vector<std::vector<int32_t>> out;
        std::vector<Iv> cur;
        auto flush = [&]() {
            if (cur.empty()) return;
            bool ok = true;
            for (size_t i = 0; i < cur.size() && ok; i++)
                for (size_t j = i + 1; j < cur.size() && ok; j++)
                    ok = cur[i].h != cur[j].h && cur[i].lo <= cur[j].hi && cur[j].lo <= cur[i].hi;
            if (ok) { out.emplace_back(); for (const Iv &x : cur) out.back().push_back(x.a); }
            else for (const Iv &x : cur) out.push_back({x.a});
            cur.clear();
        };
        int32_t reach = -1;
        for (const Iv &x : iv) {
            if (!cur.empty() && x.lo > reach) flush();
            cur.push_back(x);
            reach = std::max(reach, x.hi);
        }
        flush();
        return out;
    };

    tm.lap("minimiser -> anchors map");
    // Minimisers with two anchors on ONE walk (repeats along a haplotype) are what an additive DP counts
    // twice on sight: start the relaxation with them in S instead of discovering them by a first run.
    std::set<uint32_t> S0;
    if (!getenv("PHI_NO_S0") && dev) {
        PHICHK(phi_dev_ensure(c, c->d_flags, (size_t)std::max<int64_t>(n_ids, 1)));
        PHICHK(phi_dev_ensure(c, c->d_last_walk, (size_t)std::max<int64_t>(n_ids, 1) * 4));
        HIPCHK(hipMemsetAsync(c->d_flags.p, 0, (size_t)std::max<int64_t>(n_ids, 1), c->stream));
        HIPCHK(hipMemsetAsync(c->d_last_walk.p, 0xFF, (size_t)std::max<int64_t>(n_ids, 1) * 4, c->stream));
        {
            int64_t lo = 0;                                    // the kept anchors come in walk order: walk h has h_n_anchors[h] of them
            for (int32_t h = 0; h < nw; h++) {
                phi_launch_repeat_walk(c->stream, d_tri, lo, lo + c->h_n_anchors[h], h, c->d_last_walk.as<int32_t>(), c->d_flags.as<uint8_t>());
                lo += c->h_n_anchors[h];
            }
            if (lo != n_dp) return phi_fail(c, PHI_ERR_DEVICE, "anchors per walk do not add up (internal error)");
        }
        int64_t n_rep = 0;
        PHICHK(phi_compact(c, c->d_flags.as<uint8_t>(), n_ids, c->d_list, &n_rep));
        std::vector<uint32_t> rep((size_t)n_rep);
        if (n_rep) HIPCHK(phi_copy_sync(c, rep.data(), c->d_list.p, (size_t)n_rep * 4, hipMemcpyDeviceToHost));
        S0.insert(rep.begin(), rep.end());                     // (ascending: linear-time insertion)
        if (tm.on) fprintf(stderr, "[phi timing] solve: %zu minimisers repeat along a walk\n", S0.size());
    } else if (!getenv("PHI_NO_S0")) {
        std::vector<std::vector<uint32_t>> found(phi_host_threads());
        phi_parallel_chunks(n_ids, (int64_t)1 << 14, [&](int64_t lo, int64_t hi, int worker) {
            for (int64_t u = lo; u < hi; u++)
                for (int32_t j = sa_off[u] + 1; j < sa_off[u + 1]; j++)
                    if (dp_walk[sa_idx[j]] == dp_walk[sa_idx[j - 1]]) { found[worker].push_back((uint32_t)u); break; }
        });
        for (const auto &f : found) S0.insert(f.begin(), f.end());
        if (tm.on) fprintf(stderr, "[phi timing] solve: %zu minimisers repeat along a walk\n", S0.size());
    }
    tm.lap("repeat set");
    DpHost H;
    std::vector<uint8_t> wgt;                                  // host weights (host mode)
    if (!dev) wgt.assign((size_t)n_dp, 1);
    std::vector<Seg> best_segs;
    int64_t incumbent = INT64_MIN, global_ub = INT64_MIN, best_cov = 0;
    int n_runs = 0;
    std::vector<int32_t> cov_all, cov_w;                       // per minimiser: dp anchors traversed (all / weighted)
    std::vector<uint32_t> touched;                             // minimisers with cov_all > 0
    if (dev) {
        PHICHK(phi_dev_ensure(c, c->d_cov_all, (size_t)std::max<int64_t>(n_ids, 1) * 4));
        PHICHK(phi_dev_ensure(c, c->d_cov_w, (size_t)std::max<int64_t>(n_ids, 1) * 4));
        HIPCHK(hipMemsetAsync(c->d_cov_all.p, 0, (size_t)std::max<int64_t>(n_ids, 1) * 4, c->stream));
        HIPCHK(hipMemsetAsync(c->d_cov_w.p, 0, (size_t)std::max<int64_t>(n_ids, 1) * 4, c->stream));
    } else { cov_all.assign((size_t)n_ids, 0); cov_w.assign((size_t)n_ids, 0); }
    // device mode ends where the branch and bound proper begins: the host copies arrive, the search goes on unchanged
    auto to_host = [&]() -> int {
        PHICHK(ensure_host_walks(c));                           // (clusters_of indexes the walk entries freely)
        if (!dev) return PHI_OK;
        PhiStageTimer th("solve");
        PHICHK(phi_host_anchors(c));
        if (n_dp) PHICHK(build_csr(true));                      // (the minimiser -> anchors map, made now that it is needed)
        else { sa_off.assign((size_t)n_ids + 1, 0); sa_idx.assign(1, 0); }
        wgt.assign((size_t)n_dp, 1);
        cov_all.assign((size_t)n_ids, 0); cov_w.assign((size_t)n_ids, 0);
        touched.clear();
        dev = false;
        th.lap("  anchors and their map to the host (branch and bound)");
        return PHI_OK;
    };
    // branch and bound is finite and exact.  The reference's model.optimize() (ILP_index.cpp:1412-1418) has no
    // limit; this search has a budget counted in DP runs (phi_set_solve_budget, default 4096; <= 0 = none), never
    // in wall-clock time: the same input gives the same `optimal` flag on every run and every machine.
    const int64_t max_runs = c->solve_budget;
    auto out_of_runs = [&]() { return max_runs > 0 && n_runs >= max_runs; };
    bool exhausted = false;
    std::vector<Node> stack;
    stack.push_back(Node{});
    std::vector<int64_t> open_ub;                              // bounds of nodes given up on
    while (!stack.empty()) {
        Node node = stack.back();
        stack.pop_back();
        std::map<uint32_t, int32_t> assign(node.assign.begin(), node.assign.end());
        std::map<uint32_t, std::vector<std::vector<int32_t>>> assign_clusters;
        if (!assign.empty()) PHICHK(to_host());
        for (auto &kv : assign) assign_clusters[kv.first] = clusters_of(kv.first);
        std::set<uint32_t> S;
        if (node.assign.empty()) S = S0;
        std::set<std::set<uint32_t>> seenS;
        bool closed = false;
        int64_t node_ub = node.ub;
        uint32_t branch_slot = 0;
        bool have_branch = false, have_sets = false;
        std::set<uint32_t> lastD, lastZ;
        for (int iter = 0; iter < 8 && !closed; iter++) {
            if (out_of_runs()) { exhausted = true; break; }
            // weights of this relaxation
            std::vector<uint32_t> Sv;                          // S as a list (device mode)
            if (dev) {
                // (no assignments in device mode: to_host() runs before the first node that has one)
                Sv.assign(S.begin(), S.end());
                PHICHK(phi_dev_ensure(c, c->d_slots, std::max<size_t>(Sv.size(), 1) * 4));
                PHICHK(phi_dev_ensure(c, c->d_slots2, std::max<size_t>(Sv.size(), 1) * 4));
                PHICHK(phi_dev_ensure(c, c->d_in_s, (size_t)std::max<int64_t>(n_ids, 1)));
                if (!Sv.empty()) HIPCHK(hipMemcpyAsync(c->d_slots.p, Sv.data(), Sv.size() * 4, hipMemcpyHostToDevice, c->stream));
                // a flag per minimiser of S, then one pass over the anchors (no minimiser -> anchors map in device mode)
                HIPCHK(hipMemsetAsync(c->d_in_s.p, 0, (size_t)std::max<int64_t>(n_ids, 1), c->stream));
                phi_launch_mark_list(c->stream, c->d_slots.as<int32_t>(), nullptr, (int64_t)Sv.size(), c->d_in_s.as<uint8_t>());
                phi_launch_weights(c->stream, d_tri, n_dp, c->d_in_s.as<uint8_t>(), c->d_a_weight.as<uint8_t>());
                HIPCHK(hipStreamSynchronize(c->stream));       // Sv is read by the copy
            } else {
                std::fill(wgt.begin(), wgt.end(), 1);
                for (uint32_t s : S) for (int32_t a : anchors_of(s)) wgt[a] = 0;
                for (auto &kv : assign) {
                    if (S.count(kv.first)) continue;
                    const auto &cl = assign_clusters[kv.first];
                    for (size_t ci = 0; ci < cl.size(); ci++)
                        if ((int32_t)ci != kv.second) for (int32_t a : cl[ci]) wgt[a] = 0;
                }
            }
            int64_t val = 0;
            std::vector<Seg> segs;
            tm.lap("  weights of the relaxation");
            PHICHK(run_dp(c, dev ? nullptr : &wgt, H, &val, &segs));
            n_runs++;
            tm.lap("  DP run + backtrack");
            // exact value of this path and its additive value under wgt
            const int64_t n_sw = (int64_t)segs.size() - 1;
            int64_t add_w = 0, n_touched = 0;
            std::set<uint32_t> D, Z;
            if (dev) {
                // cover counts per minimiser on the device; back come the two sums and the two short lists
                std::vector<phi_ent_t> sg(segs.size() * 2);
                for (size_t i = 0; i < segs.size(); i++) { sg[2 * i] = (phi_ent_t)segs[i].es; sg[2 * i + 1] = (phi_ent_t)segs[i].ee; }
                const int64_t twice_cap = (int64_t)1 << 22;
                PHICHK(phi_dev_ensure(c, c->d_segs, std::max<size_t>(sg.size(), 2) * 4));
                PHICHK(phi_dev_ensure(c, c->d_list, (size_t)twice_cap * 4));
                unsigned long long *d_ctr = c->d_ctr.as<unsigned long long>();
                HIPCHK(hipMemcpyAsync(c->d_segs.p, sg.data(), sg.size() * 4, hipMemcpyHostToDevice, c->stream));
                HIPCHK(hipMemsetAsync(d_ctr, 0, 32, c->stream));
                phi_launch_path_cover(c->stream, false, c->d_segs.as<phi_ent_t>(), (int32_t)segs.size(), c->d_g_off.as<int64_t>(), d_tri, c->d_a_weight.as<uint8_t>(),
                                      c->d_cov_all.as<int32_t>(), c->d_cov_w.as<int32_t>(), d_ctr, c->d_list.as<uint32_t>(), twice_cap);
                phi_launch_uncovered_slots(c->stream, c->d_slots.as<uint32_t>(), (int64_t)Sv.size(), c->d_cov_all.as<int32_t>(), d_ctr, c->d_slots2.as<uint32_t>());
                phi_launch_path_cover(c->stream, true, c->d_segs.as<phi_ent_t>(), (int32_t)segs.size(), c->d_g_off.as<int64_t>(), d_tri, c->d_a_weight.as<uint8_t>(),
                                      c->d_cov_all.as<int32_t>(), c->d_cov_w.as<int32_t>(), d_ctr, c->d_list.as<uint32_t>(), twice_cap);
                unsigned long long hc[4];
                HIPCHK(hipMemcpyAsync(hc, d_ctr, 32, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                if ((int64_t)hc[2] > twice_cap) return phi_fail(c, PHI_ERR_OVERFLOW, "more than 2^22 minimisers counted twice by one path");
                add_w = (int64_t)hc[0]; n_touched = (int64_t)hc[1];
                std::vector<uint32_t> dl((size_t)hc[2]), zl((size_t)hc[3]);
                if (hc[2]) HIPCHK(phi_copy_sync(c, dl.data(), c->d_list.p, dl.size() * 4, hipMemcpyDeviceToHost));
                if (hc[3]) HIPCHK(phi_copy_sync(c, zl.data(), c->d_slots2.p, zl.size() * 4, hipMemcpyDeviceToHost));
                D.insert(dl.begin(), dl.end());
                Z.insert(zl.begin(), zl.end());
            } else {
                for (uint32_t s : touched) { cov_all[s] = 0; cov_w[s] = 0; }
                touched.clear();
                for_covered(c, segs, [&](int64_t a) {
                    const uint32_t s = c->h_dp[a].slot;
                    if (!cov_all[s]++) touched.push_back(s);
                    if (wgt[a]) { cov_w[s]++; add_w++; }
                });
                n_touched = (int64_t)touched.size();
            }
            add_w -= cost * n_sw;
            if (add_w != val) return phi_fail(c, PHI_ERR_DEVICE, "DP value %lld != value %lld of its own path (internal error)", (long long)val, (long long)add_w);
            const int64_t true_val = n_touched - cost * n_sw;
            if (true_val > incumbent) { incumbent = true_val; best_segs = segs; best_cov = n_touched; }
            const int64_t ub = val + (int64_t)S.size();
            node_ub = std::min(node_ub, ub);
            if (n_runs == 1) global_ub = ub;
            // a search that goes on says so (the reference's model.optimize() prints Gurobi's log; the command line runs without a
            // budget unless --dp-budget gives one): every 64 DP runs, and the 16th (PHI_PROGRESS=0: never; =n: every n runs)
            {
                static const int every = getenv("PHI_PROGRESS") ? atoi(getenv("PHI_PROGRESS")) : 64;
                if (every > 0 && (n_runs % every == 0 || (every == 64 && n_runs == 16)))
                    fprintf(stderr, "[M::solve] exact search: %d DP runs, best path %lld, proven bound %lld, %zu open node(s)%s\n", n_runs,
                            (long long)incumbent, (long long)global_ub, stack.size() + 1, max_runs > 0 ? "" : " (no budget: --dp-budget N limits it)");
            }
            if (node_ub <= incumbent) { closed = true; break; }
            // tighten: bound doubly-counted minimisers by the constant 1, release unused constants
            if (!dev) for (uint32_t s : touched) if (cov_w[s] >= 2) D.insert(s);
            if (!dev) for (uint32_t s : S) {
                // is any anchor this node still allows for s traversed?
                bool covered = false;
                if (cov_all[s]) {
                    auto it = assign.find(s);
                    if (it == assign.end()) covered = true;
                    else {
                        const auto &cl = assign_clusters[s][it->second];
                        for_covered(c, segs, [&](int64_t a) {
                            if (c->h_dp[a].slot == s && std::find(cl.begin(), cl.end(), (int32_t)a) != cl.end()) covered = true;
                        });
                    }
                }
                if (!covered) Z.insert(s);
            }
            if (tm.on) fprintf(stderr, "[phi timing] solve:   run %d: DP value %lld + |S| %zu = bound %lld; path: exact %lld, %lld switches; doubly counted %zu, unused constants %zu\n",
                               n_runs, (long long)val, S.size(), (long long)ub, (long long)true_val, (long long)n_sw, D.size(), Z.size());
            tm.lap("  path value, tighten sets");
            if (D.empty() && Z.empty()) { closed = true; break; }   // bound attained by this path
            // remember what a branching candidate is chosen from (below, in canonical = first anchor order)
            lastD = D; lastZ = Z; have_sets = true;
            std::set<uint32_t> S2 = S;
            for (uint32_t s : D) S2.insert(s);
            for (uint32_t s : Z) S2.erase(s);
            if (S2 == S || seenS.count(S2)) break;
            seenS.insert(S);
            S = S2;
        }
        if (closed) continue;
        if (have_sets && !exhausted) {
            PHICHK(to_host());                                 // (the minimiser -> anchors map)
            int32_t best_a = INT32_MAX;
            for (uint32_t s : lastD) if (anchors_of(s)[0] < best_a) { best_a = anchors_of(s)[0]; branch_slot = s; }
            if (lastD.empty()) for (uint32_t s : lastZ) if (!assign.count(s) && anchors_of(s)[0] < best_a) { best_a = anchors_of(s)[0]; branch_slot = s; }
            have_branch = best_a != INT32_MAX;
        }
        if (exhausted || !have_branch) { open_ub.push_back(std::min(node_ub, global_ub)); if (exhausted) break; continue; }
        const auto cl = clusters_of(branch_slot);
        for (int32_t ci = (int32_t)cl.size() - 1; ci >= 0; ci--) {
            Node ch = node;
            ch.ub = node_ub;
            ch.assign.emplace_back(branch_slot, ci);
            stack.push_back(ch);
        }
    }
    for (const Node &n : stack) open_ub.push_back(std::min(n.ub, global_ub));
    int64_t ub = incumbent;
    for (int64_t u : open_ub) ub = std::max(ub, u);

    tm.lap("DP runs + certificate");
    // ---- 5. decode (:1431-1525)
    int64_t n_path_vtx = 0;
    for (const Seg &s : best_segs) n_path_vtx += s.ee - s.es + 1;
    c->h_path_vtx.resize((size_t)n_path_vtx);
    c->h_path_hap.resize((size_t)n_path_vtx);
    int64_t hap_len = 0;
    {
        int64_t o = 0;
        for (const Seg &s : best_segs) {
            PHICHK(walk_vtx_range(c, s.es, s.ee - s.es + 1, c->h_path_vtx.data() + o));
            std::fill(c->h_path_hap.begin() + o, c->h_path_hap.begin() + o + (s.ee - s.es + 1), s.h);
            for (int64_t i = 0; i <= s.ee - s.es; i++) { const int32_t v = c->h_path_vtx[(size_t)(o + i)]; hap_len += c->h_seq_off[v + 1] - c->h_seq_off[v]; }
            o += s.ee - s.es + 1;
        }
    }
    // adjacent label changes (:1517-1519): the labels change exactly between the path's stretches
    int32_t recomb = 0;
    for (size_t i = 1; i < best_segs.size(); i++) recomb += best_segs[i].h != best_segs[i - 1].h;
    const int64_t n_cov = best_cov;                            // minimisers covered by the best path (counted when it was found)

    phi_result &R = c->result;
    R.objective = incumbent;
    R.upper_bound = ub;
    R.optimal = ub == incumbent;
    R.n_dp_runs = n_runs;
    R.n_covered = n_cov;
    R.n_path = (int64_t)c->h_path_vtx.size();
    R.path_vtx = c->h_path_vtx.data();
    R.path_hap = c->h_path_hap.data();
    R.recombination_count = recomb;
    R.n_switches = (int32_t)best_segs.size() - 1;
    R.hap_len = hap_len;
    R.n_walks = nw;
    R.n_minimizers = c->h_n_minimizers.data();
    R.n_anchors = c->h_n_anchors.data();
    R.spectrum_size = spectrum;
    R.filtered = filtered;
    R.retained = spectrum - filtered;
    R.n_in_model = in_model;
    tm.lap("decode");
    c->solved = true;
    return PHI_OK;
}
