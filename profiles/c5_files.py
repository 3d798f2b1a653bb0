#!/usr/bin/env python3
"""BASELINE config 5 at its stated size FROM FILES: the native generator writes the GFA (200 walks x 170 Mbp: ~12 GB of text)
and the 30x read set (34 M reads: ~10.5 GB of FASTQ) to a RAM disk, then the drop-in command line runs on them --
process start to closed FASTA, with its stage table (PHI_TIMING=1).  Not part of the default bench (20 GB of files).

    python3 profiles/c5_files.py [--config C5|C5s] [--dir /dev/shm/phi_c5] [--fasta] [--out gpurun_out/r03/c5_files.json]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C5")
    ap.add_argument("--dir", default="/dev/shm/phi_c5_files")
    ap.add_argument("--fasta", action="store_true", help="reads as FASTA instead of 4-line FASTQ")
    ap.add_argument("--runs", type=int, default=2)
    ap.add_argument("--out", default=None)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--gap", type=float, default=4.0, help="seconds between runs: the driver clears the 65 GB of HBM the last process gave back before it hands "
                    "them out again, and a process that starts within a second of the last one's end waits ~1 s in one of its large allocations "
                    "(r04i: PHI_TIMING_ALLOC=1 shows one hipMalloc of 11.4 GB taking 0.26 ms or 1 s, in alternate runs 1 s apart)")
    args = ap.parse_args()
    sys.path.insert(0, ROOT)
    import bench
    from phi_amd import synth
    gk, s_seed, n_mosaic, r_seed, cov = synth.NATIVE_CONFIGS[args.config]
    os.makedirs(args.dir, exist_ok=True)
    out = {"config": args.config, "dir": args.dir}
    try:
        t0 = time.perf_counter()
        g = synth.NativeGraph(**gk)
        truth = g.sample(s_seed, n_mosaic)
        out["generate_graph_s"] = time.perf_counter() - t0
        gfa = os.path.join(args.dir, "g.gfa")
        rd = os.path.join(args.dir, "r.fa" if args.fasta else "r.fq")
        fa = os.path.join(args.dir, "out.fa")
        t0 = time.perf_counter()
        out["gfa_bytes"] = g.write_gfa(gfa)
        out["write_gfa_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        n_reads = g.n_reads(cov)
        out["reads_bytes"] = g.write_reads(rd, r_seed, 0, n_reads, fastq=not args.fasta)
        out["write_reads_s"] = time.perf_counter() - t0
        out["n_reads"], out["n_walks"], out["n_vtx"], out["walk_entries"] = n_reads, g.n_walks, g.n_vtx, int(g.walk_off[-1])
        out["truth_walks"] = truth["walks"]
        g.close()
        del g
        print(json.dumps(out), flush=True)
        phi = os.path.join(ROOT, "phi_amd", "PHI")
        runs = []
        for i in range(args.runs):
            time.sleep(args.gap)
            t_spawn = time.time()
            t0 = time.perf_counter()
            r = subprocess.run([phi, "-g", gfa, "-r", rd, "-o", fa], capture_output=True, text=True, env=dict(os.environ, PHI_TIMING="1"))
            dt = time.perf_counter() - t0
            stages, info = bench._phi_stage_table(r.stderr)
            log = [l for l in r.stderr.splitlines() if not l.startswith("syn")]
            rec = {"rc": r.returncode, "wall_s": dt, "spawn_to_fasta_closed_s": info.get("fasta_closed_epoch", t_spawn + dt) - t_spawn,
                   "stages_s": {k: v for k, v in stages.items() if k != "detail"}, "detail_ms": stages.get("detail", {}),
                   "log": [l for l in log if not l.startswith("[phi timing] main: stage")][-160:]}
            for l in log:
                if "Real time" in l:
                    rec["phi_line"] = l
                if "resident now" in l:
                    rec["resident"] = l.split("main: ", 1)[1]
                if l.startswith("Recombined haplotypes"):
                    rec["recombined"] = l[:300]
            runs.append(rec)
            print(json.dumps({k: rec.get(k) for k in ("rc", "wall_s", "spawn_to_fasta_closed_s", "phi_line", "resident", "stages_s")}), flush=True)
            if r.returncode != 0:
                print(r.stderr[-3000:])
                break
        out["runs"] = runs
        if os.path.exists(fa):
            out["fasta_bytes"] = os.path.getsize(fa)
    finally:
        if not args.keep:
            shutil.rmtree(args.dir, ignore_errors=True)
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "runs"}))


if __name__ == "__main__":
    main()
