#!/usr/bin/env python3
"""bench.py -- headline benchmark of the PHI hot path on MI355X.

Metric (BASELINE.json): read Gbases/s scored on the 49-haplotype MHC graph, plus end-to-end
seconds.  The real graph cannot be built offline, so the workload is SURVEY.md 8(d)'s
deterministic stand-in "synMHC-49" (config C2: 49 walks x ~5.15 Mbp, nodes <= 30 bp, 1x 150-bp
reads).  One STEP = one scoring pass of the read batch, reads already resident in HBM:
    reset spectrum/hits -> 2-bit pack -> (w,k)-minimiser sketch + murmur3 -> spectrum insert +
    probe of the walk-minimiser table   [+ RCCL all-reduce(MAX) of the hit vector when N > 1]
The graph index (walk sketch + table) is built once before the timed region and the exact solve
(filter + DP + certificate) runs once after it; both are timed and reported separately, and
`end_to_end_s` = index build + one step + solve.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU; every rank holds the full graph index and its own shard of reads
(weak scaling: each rank scores one C2-sized read set per step, drawn with its own seed).  The path
has ONE exchange per job: the hit vector (one byte per distinct walk minimiser) is
a running OR over the batches a rank has scored, so it is all-reduced (MAX, RCCL over xGMI) once,
after the rank's last batch -- inside the timed region, after the K steps.  No collective per step.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(reads_concat, read_off, k, w, budget_s=12.0):
    """Oracle (scalar string-based port of compute_hashes) on the host cores: checker code timed as
    the CPU baseline, never part of the measured GPU path."""
    from oracle import oracle as O
    L = O.lib()
    raw = reads_concat.tobytes()
    n = len(read_off) - 1
    t0 = time.perf_counter()
    done = 0
    r = 0
    while True:
        a, b = int(read_off[r]), int(read_off[r + 1])
        L.orc_sketch(raw[a:b], b - a, k, w, None, None, 0)
        done += b - a
        r = (r + 1) % n
        if (r & 1023) == 0 and time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt / 1e9, "unit": "Gbases/s", "cores": 1, "kind": "port",
            "sample": f"{done / 1e6:.1f} Mbases of the same 150-bp synthetic reads, oracle/phi_oracle.c orc_sketch, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C2", help="workload of phi_amd.synth.CONFIGS, or C1syn (reference MHC_4 graph + generator reads)")
    ap.add_argument("--prof-period", type=int, default=8,
                    help="every n-th sketch launch of the timed region carries the HIP events of roofline.kernel_avg_ms "
                         "(a bracketed launch costs the stream ~5 us more than a plain one; 1 = every launch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-solve", action="store_true")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="N > 1 on ONE GPU for testing: every rank uses cuda:0 and the exchange goes through gloo on the host")
    args = ap.parse_args()

    import torch
    import __graft_entry__
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        __graft_entry__.ensure_built()                         # a checkout without built libraries (no fallback exists)
    else:
        t_wait = time.time()
        libs = [os.path.join(ROOT, "phi_amd", n) for n in ("libphi_amd.so", "libphi_host.so")]
        while not all(os.path.exists(p) for p in libs) and time.time() - t_wait < 600:
            time.sleep(1.0)                                    # rank 0 is building
    import phi_amd
    from phi_amd import dist as pdist
    from phi_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch through torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.rehearse_gloo:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group("gloo" if args.rehearse_gloo else "nccl", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    K, W = 31, 25
    t0 = time.perf_counter()
    if args.config == "C1syn":
        # the reference's own graph (test/MHC_4.gfa.gz, 5 walks) with the generator's reads
        gk = dict(seed="tests/golden/data/MHC_4.gfa.gz")
        g = synth.graph_from_gfa(os.path.join(ROOT, "tests", "golden", "data", "MHC_4.gfa.gz"))
        rk = dict(coverage=1.0, seed=4102, n_mosaic=2)
    else:
        gk, rk = synth.CONFIGS[args.config]
        g = synth.make_graph(**gk)
    rk = dict(rk)
    if rank:
        rk["sample_seed"] = rk["seed"] + 1000 * rank           # each rank scores its own reads of the same sample
    bases, off, truth = synth.make_reads(g, **rk)
    t_gen = time.perf_counter() - t0
    n_reads, n_bases = len(off) - 1, int(off[-1])

    # one explicit stream for everything: the kernels behind the C ABI, torch's copies and the RCCL
    # collective are ordered on it (torch's default stream is the null stream, which the library
    # would replace by a private non-blocking one -- unordered against torch)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = phi_amd.Context(local_rank)
    ctx.set_params(k=K, w=W, threshold=1.0, recombination=100)
    ctx.set_stream(stream.cuda_stream)
    A = g.arrays()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], A["top_rank"])
    torch.cuda.synchronize()
    t_index = time.perf_counter() - t0
    walk_bases = int((A["seq_off"][A["walk_vtx"] + 1] - A["seq_off"][A["walk_vtx"]]).sum())

    d_bases = torch.from_numpy(bases).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    def step():
        ctx.reset_reads()
        ctx.add_reads_device(d_bases.data_ptr(), d_off.data_ptr(), n_reads, n_bases)

    def exchange():
        # the job's one data-path collective: all-reduce (MAX) of the hit vector, in place
        if world > 1:
            hit_ptr, n_unique = ctx.hits_buffer()
            pdist.allreduce_hits(torch.as_tensor(pdist.DevArray(hit_ptr, n_unique), device=dev))

    # clock ramp: a fresh box runs its first launches at idle clocks (a third slower for the first tens
    # of milliseconds); untimed, before the W warmup steps
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.25:
        for _ in range(50):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    exchange()                                                 # warms RCCL up as well
    ctx.prof_read()                                            # drop warmup timings
    ctx.prof_enable(args.prof_period)                               # every n-th sketch launch carries timing events
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    exchange()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ctx.prof_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_gloo else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_launch, kern_ms, kern_bases = ctx.prof_read()
    stats = ctx.reads_stats()
    density = stats["n_emitted"] / max(1, stats["n_bases"])

    # ---- the rest of the job, once: spectrum merge (N > 1), filter + exact solve
    t_solve = None
    res = None
    if not args.no_solve:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pdist.merge_spectrum_into(ctx, dev)
        res = ctx.solve()
        torch.cuda.synchronize()
        t_solve = time.perf_counter() - t0
        if world > 1:
            o = torch.tensor([res["objective"], -res["objective"]], dtype=torch.int64, device="cpu" if args.rehearse_gloo else dev)
            dist.all_reduce(o, op=dist.ReduceOp.MAX)
            assert int(o[0].item()) == -int(o[1].item()), "ranks disagree on the objective"

    ms_per_step = elapsed / args.steps * 1e3
    value = n_bases * world * args.steps / elapsed / 1e9
    b_alg = 1.5 + 24.0 * density
    kern_avg_ms = kern_ms / max(1, n_launch)
    achieved = (kern_bases / max(1, n_launch)) * b_alg / (kern_avg_ms * 1e-3) / 1e9 if n_launch else 0.0

    # HBM bytes per launch from PMC counters: measured in a separate rocprofv3 run (bench.py cannot
    # profile itself) and kept, with the commands, under profiles/
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if tj.get("config") == args.config:
            traffic = tj["bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass

    out = {
        "metric": "read Gbases/s scored (49-hap MHC graph stand-in synMHC-49)",
        "value": value, "unit": "Gbases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"{args.config}: {'reference graph' if args.config == 'C1syn' else f'synMHC-{g.n_walks} graph'} (seed {gk['seed']}, {g.n_walks} walks, {g.n_vtx} vertices{'' if args.config == 'C1syn' else ' <=30 bp'}, "
                               f"{walk_bases / 1e6:.1f} Mbases of walks) + {n_reads} reads of mean {n_bases / max(1, n_reads):.0f} bp per GPU (seed {rk['seed']}, {n_bases / 1e6:.2f} Mbases)",
                   "k": K, "w": W, "R": 100, "reads_per_gpu_bases": n_bases, "parallelism": f"read-shard x{world}"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "phi_sketch_kernel<PHI_MODE_PROBE>", "kernel_avg_ms": kern_avg_ms,
                     "kernel_launches": n_launch, "kernel_timed_every": args.prof_period, "bytes_per_base_algorithmic": b_alg, "minimiser_density": density,
                     "kernel_gbases_per_s": (kern_bases / max(1, n_launch)) / (kern_avg_ms * 1e-3) / 1e9 if n_launch else 0.0},
        "index_build_s": t_index, "graph_gbases_per_s": walk_bases / t_index / 1e9,
        "solve_s": t_solve, "end_to_end_s": (t_index + ms_per_step * 1e-3 + t_solve) if t_solve is not None else None,
        "synthetic_gen_s": t_gen,
    }
    if res is not None:
        out["result"] = {k_: int(res[k_]) for k_ in ("objective", "upper_bound", "optimal", "n_dp_runs", "n_covered",
                                                      "recombination_count", "n_switches", "spectrum_size", "filtered",
                                                      "n_in_model", "hap_len")}
        out["result"]["truth_walks"] = truth["walks"]
        out["result"]["path_walks"] = [int(x) for x in res["path_hap"][np.r_[True, res["path_hap"][1:] != res["path_hap"][:-1]]]]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(bases, off, K, W)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
