"""A graph with more than 2^31 walk entries, solved and property-checked (not part of the test suite: ~120 GB of HBM).

Walk entries are 32-bit UNSIGNED indices below PHI_MAX_ENTRIES = 2^32 - 64 (phi_kernels.h: phi_ent_t); until round 3 they
were signed and the library refused 2^31 entries.  This script builds, with the native generator (phi_amd/synth.py
NativeGraph), a chromosome-1-scale graph -- by default 256 walks over a 250 Mbp backbone, 2.5 G walk entries, the
last ~35 walks lie above entry 2^31 -- samples a mosaic that crosses walks on BOTH sides of 2^31, adds 30x reads and
checks what tests/test_gpu_parity.py::test_chromosome_scale_properties checks for C5:
  * the solve proves its path optimal (objective == upper bound),
  * the generator's truth walks come back, in order,
  * the path's objective, recounted in numpy from the kept anchors the library reports, equals the reported one,
  * anchors per walk equal a bincount of the kept anchors,
  * the per-walk minimisers of the de-duplicated index equal a direct sketch of the walk's sequence -- for the LAST walk
    (entries above 2^31) and for walk 0.

    python3 profiles/wide_entries.py [--walks 256] [--backbone 250000000] [--coverage 30] [--out gpurun_out/r03/wide_entries.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def mem_available_gb():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            return int(line.split()[1]) / 1e6
    return 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--walks", type=int, default=256)
    ap.add_argument("--backbone", type=int, default=250_000_000)
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--mosaic", type=int, default=3)
    ap.add_argument("--seed", type=int, default=20001)
    ap.add_argument("--batch-reads", type=int, default=8_000_000)
    ap.add_argument("--no-recount", action="store_true", help="skip the numpy recount from the kept anchors (tens of GB of host memory)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()

    import __graft_entry__
    __graft_entry__.ensure_built()
    import phi_amd
    from phi_amd import synth
    from test_gpu_parity import _evaluate_path_numpy

    out = dict(walks=a.walks, backbone=a.backbone, coverage=a.coverage, host_mem_available_gb=round(mem_available_gb(), 1))
    t0 = time.time()
    g = synth.NativeGraph(backbone_len=a.backbone, n_walks=a.walks, seed=a.seed)
    A = g.arrays()
    n_entries = int(A["walk_off"][-1])
    out.update(n_vtx=int(g.n_vtx), n_entries=n_entries, generate_s=round(time.time() - t0, 2))
    print(f"[wide] graph: {g.n_vtx} vertices, {a.walks} walks, {n_entries} walk entries ({n_entries / 2**31:.3f} x 2^31) in {out['generate_s']} s", flush=True)
    assert n_entries > 2**31, "this script is about graphs beyond 2^31 walk entries: raise --walks or --backbone"
    first_high = int(np.searchsorted(A["walk_off"], 2**31, side="right")) - 1          # the walk that holds entry 2^31
    # a mosaic with walks on both sides of entry 2^31 (the first sample seed that gives one)
    truth = None
    for s_seed in range(a.seed + 1, a.seed + 400):
        t = g.sample(s_seed, a.mosaic)
        if max(t["walks"]) > first_high and min(t["walks"]) < first_high:
            truth = t
            break
    assert truth is not None, "no sample seed crosses 2^31"
    out.update(sample_seed=s_seed, truth_walks=truth["walks"], first_walk_above_2_31=first_high + 1)
    print(f"[wide] truth walks {truth['walks']} (walks > {first_high} lie above entry 2^31)", flush=True)

    ctx = phi_amd.Context(0)
    ctx.set_params(k=31, w=25, threshold=1.0, recombination=100)
    t1 = time.time()
    ctx.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], A["top_rank"])
    out["set_graph_s"] = round(time.time() - t1, 2)
    print(f"[wide] set_graph {out['set_graph_s']} s", flush=True)
    n = g.n_reads(a.coverage)
    t2 = time.time()
    gen = add = 0.0
    for lo in range(0, n, a.batch_reads):
        hi = min(n, lo + a.batch_reads)
        tg = time.time()
        b, o = g.reads(a.seed + 2, lo, hi)
        gen += time.time() - tg
        ta = time.time()
        ctx.add_reads((b, o))
        add += time.time() - ta
        del b, o
    out.update(n_reads=int(n), read_bases=int(n) * 150, reads_generate_s=round(gen, 2), add_reads_s=round(add, 2))
    print(f"[wide] {n} reads: generated in {gen:.1f} s, added in {add:.1f} s", flush=True)
    t3 = time.time()
    res = ctx.solve()
    out["solve_s"] = round(time.time() - t3, 2)
    info = ctx.index_stats()
    hap = res["path_hap"]
    walks = [int(x) for x in hap[np.r_[True, hap[1:] != hap[:-1]]]]
    out.update(objective=int(res["objective"]), upper_bound=int(res["upper_bound"]), optimal=int(res["optimal"]), n_dp_runs=int(res["n_dp_runs"]),
               n_covered=int(res["n_covered"]), n_switches=int(res["n_switches"]), path_walks=walks, n_classes=int(info["n_classes"]),
               n_walk_minimizers=int(info["n_walk_minimizers"]), n_anchors=int(res["n_anchors"].sum()), hap_len=int(res["hap_len"]))
    print(f"[wide] solve {out['solve_s']} s: objective {out['objective']} (bound {out['upper_bound']}), path walks {walks}, {out['n_anchors']} anchors, {out['n_dp_runs']} DP runs", flush=True)
    assert info["n_entries"] == n_entries and info["n_classes"] < n_entries // 20
    assert info["n_walk_minimizers"] == int(res["n_minimizers"].sum())
    assert res["optimal"] == 1 and res["objective"] == res["upper_bound"]
    assert walks == truth["walks"], (walks, truth["walks"])
    out["checks"] = ["optimal", "truth walks recovered"]
    # per-walk minimisers of the de-duplicated index against a direct sketch of the walk's sequence
    for h in (g.n_walks - 1, 0):
        wh, wp = ctx.walk_minimizers(h)
        sh, sp, _ = ctx.sketch([g.walk_sequence(h).tobytes()], 31, 25)
        assert np.array_equal(wh, sh) and np.array_equal(wp, sp) and len(wh) == res["n_minimizers"][h], h
        del wh, wp, sh, sp
    out["checks"].append(f"walk minimisers == direct sketch (walks {g.n_walks - 1}, 0)")
    print("[wide] walk minimisers of the last and the first walk equal the direct sketch", flush=True)
    need_gb = out["n_anchors"] * 40 / 1e9
    if a.no_recount or mem_available_gb() < need_gb + 16:
        out["recount"] = f"skipped ({mem_available_gb():.0f} GB available, ~{need_gb:.0f} GB needed)" if not a.no_recount else "skipped (--no-recount)"
    else:
        t4 = time.time()
        kept = ctx.kept_anchors()
        obj, n_cov, n_sw = _evaluate_path_numpy(A, res, kept, 100)
        assert (obj, n_cov, n_sw) == (res["objective"], res["n_covered"], res["n_switches"]), (obj, n_cov, n_sw)
        assert res["n_anchors"].tolist() == np.bincount(kept[1], minlength=g.n_walks).tolist()
        del kept
        out["recount_s"] = round(time.time() - t4, 2)
        out["checks"] += ["numpy recount of the path's objective from the kept anchors", "anchors per walk == bincount of the kept anchors"]
        print(f"[wide] numpy recount from the kept anchors agrees ({out['recount_s']} s)", flush=True)
    import torch
    out["hbm_peak_gb"] = None
    try:
        free, total = torch.cuda.mem_get_info(0)
        out["hbm_in_use_at_end_gb"] = round((total - free) / 1e9, 1)
    except Exception:
        pass
    ctx.close()
    out["total_s"] = round(time.time() - t0, 1)
    line = json.dumps(out)
    print(line, flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(line + "\n")


if __name__ == "__main__":
    main()
